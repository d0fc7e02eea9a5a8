cd $GRAFT_REPO_ROOT
O=gpurun_out/c69; mkdir -p $O
python bench.py --input text --no-cpu-baseline --steps 5 > $O/b_text_1.json 2> $O/e1
python bench.py --no-cpu-baseline --steps 5 > $O/b_zipf_1.json 2> $O/e2
python bench.py --mode decompress --level 5 --input mixed --frame-mib 1 --no-cpu-baseline > $O/d_1m.json 2> $O/e3
python bench.py --input text --size-mib 10 --steps 50 --no-cpu-baseline > $O/b_text_10m.json 2> $O/e4
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c69/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['value'], d.get('compress_MBps_per_gpu'), d.get('decompress_MBps_per_gpu'), {k.split('/')[1]:round(v,2) for k,v in d['stage_ms'].items() if v>0.25})
PY
