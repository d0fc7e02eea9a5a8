#!/bin/bash
# SQ counter passes (8 SQ slots per pass, MI355X_MICROARCH.md "rocprofv3 PMC slots") over one bench workload, per-kernel sums.
# usage: BENCH_ARGS="..." bash tools/pmc_sq.sh <tag> ; writes gpurun_out/prof/<tag>_sq.json
TAG=${1:-sq}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS}"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
P3="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT"
for i in 1 2 3; do
  eval "P=\$P$i"
  rm -rf "$OUT/sq$i"
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/sq$i" -o sq -- python3 "$ROOT/bench.py" $ARGS > "$OUT/sq$i.out" 2> "$OUT/sq$i.err" || echo "pass $i failed: $(tail -2 $OUT/sq$i.err)"
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
res = {}
for i in (1, 2, 3):
    for path in glob.glob(os.path.join(out, f"sq{i}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("zmi::", "").strip()
            if k.startswith("at::") or k.startswith("__amd"): continue
            d = res.setdefault(k, {})
            a = d.setdefault(r["Counter_Name"], [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
final = {k: {c: v[0] / v[1] for c, v in d.items()} for k, d in res.items()}
json.dump(final, open(os.path.join(out, f"{tag}_sq.json"), "w"), indent=1)
for k, d in final.items():
    wc = d.get("SQ_WAVE_CYCLES", 0)
    if wc < 1e6: continue
    print(k[:44], " ".join(f"{c[3:]}={v/wc:.3f}" if c.startswith("SQ_WAIT") or c.startswith("SQ_ACTIVE") else f"{c[3:]}={v:.3g}" for c, v in sorted(d.items())))
PY
rm -rf "$OUT/sq1" "$OUT/sq2" "$OUT/sq3"
