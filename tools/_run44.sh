cd $GRAFT_REPO_ROOT
O=gpurun_out/c44; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
for cfg in "text 5" "mixed 5" "text 3" "mixed 3" "text 1" "zipf 1"; do set -- $cfg
  timeout -k 10 200 python bench.py --input $1 --level $2 --no-cpu-baseline --steps 5 > $O/b_$1_$2.json 2> $O/b_$1_$2.err || echo "bench $cfg failed"
  ZSTDMI_NO_REGION=1 timeout -k 10 200 python bench.py --input $1 --level $2 --no-cpu-baseline --steps 5 > $O/n_$1_$2.json 2> $O/n_$1_$2.err || echo "bench-noregion $cfg failed"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c44/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['compress_MBps_per_gpu'], d['ratio'], round(d['stage_ms']['compress/lz_fast'],2), d['round_trip_bit_exact'])
    except Exception as e: print(f, 'ERR', e)
PY
