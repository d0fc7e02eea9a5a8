#!/bin/bash
# Round profiles: for every workload of the default bench line (headline + extra_configs), the three rocprofv3 passes of
# tools/profile_round.sh (kernel stats, FETCH_SIZE, WRITE_SIZE; one workload per process).  Run on the GPU box through gpurun; then
# copy gpurun_out/prof/<tag>_* into profiles/.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
R=${1:-r03}
cd "$ROOT"
BENCH_ARGS="" bash tools/profile_round.sh ${R}_zipf
BENCH_ARGS="--input text" bash tools/profile_round.sh ${R}_text
BENCH_ARGS="--input text --size-mib 10 --steps 20" bash tools/profile_round.sh ${R}_text10
BENCH_ARGS="--input mixed --level 5" bash tools/profile_round.sh ${R}_mixed5
BENCH_ARGS="--mode decompress --input mixed --level 5 --frame-mib 1 --size-mib 1024 --unique-mib 256" bash tools/profile_round.sh ${R}_dec5g1
BENCH_ARGS="--mode decompress --input mixed --level 5 --frame-mib 1 --size-mib 4096 --unique-mib 256" bash tools/profile_round.sh ${R}_dec5
BENCH_ARGS="--mode decompress --input mixed --level 5 --frame-mib 256 --size-mib 256 --unique-mib 256" bash tools/profile_round.sh ${R}_dec5one
ls "$ROOT/gpurun_out/prof"
