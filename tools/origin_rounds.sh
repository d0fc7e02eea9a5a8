#!/bin/bash
# per-dispatch durations of the origin path's kernels (rocprofv3 kernel trace) for one long-frame decode: tools/origin_rounds.sh kind MiB frames level
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/origin_trace; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python3 "$ROOT/tools/long_frame_time.py" "$@" > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "origin_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last decode call only: from the last origin_fill on
last = max(i for i, r in enumerate(rows) if "origin_fill" in r["Kernel_Name"])
for r in rows[last:]:
    print(r["Kernel_Name"].split("(")[0].replace("zmi::", ""), round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1), "us")
PY
