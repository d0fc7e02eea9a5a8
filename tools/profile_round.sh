#!/bin/bash
# Three rocprofv3 passes over the default bench workload (run on the GPU box through gpurun):
#   1. --kernel-trace --stats            -> per-kernel average durations
#   2. --pmc FETCH_SIZE  --kernel-trace  -> HBM read bytes per dispatch  (separate pass: TCC slots, MI355X_MICROARCH.md)
#   3. --pmc WRITE_SIZE  --kernel-trace  -> HBM write bytes per dispatch
# then tools/pmc_summary.py folds them into gpurun_out/prof/<tag>_*.{csv,json}; copy those into profiles/.
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extra ${BENCH_ARGS}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" $ARGS > "$OUT/${TAG}_bench_under_stats.json" 2> "$OUT/stats.err"
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o fetch -- python3 "$ROOT/bench.py" $ARGS > "$OUT/fetch.out" 2> "$OUT/fetch.err"
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o write -- python3 "$ROOT/bench.py" $ARGS > "$OUT/write.out" 2> "$OUT/write.err"
echo "write pass done"
python3 "$ROOT/tools/pmc_summary.py" "$OUT" "$TAG"
