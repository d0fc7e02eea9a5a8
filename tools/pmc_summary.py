#!/usr/bin/env python3
"""Fold the three rocprofv3 passes of tools/profile_round.sh into two small files:
   <tag>_kernel_stats.csv  — per-kernel calls / total / average / share (from --stats)
   <tag>_pmc_traffic.json  — per-kernel HBM bytes per dispatch from FETCH_SIZE / WRITE_SIZE.
FETCH_SIZE and WRITE_SIZE are reported in KiB-like units of 1024 B by rocprofv3's derived-metric definition; on gfx950
FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads, so it is doubled (MI355X_MICROARCH.md, HBM section).
Both raw and corrected figures are kept."""
import csv, glob, json, os, sys

out, tag = sys.argv[1], sys.argv[2]

def find(sub, suffix):
    hits = glob.glob(os.path.join(out, sub, "**", f"*{suffix}"), recursive=True)
    return hits[0] if hits else None

def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").replace("zmi::", "").strip()

stats = find("stats", "kernel_stats.csv")
rows = []
if stats:
    with open(stats) as f:
        for r in csv.DictReader(f):
            rows.append(r)
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f); w.writerow(["kernel", "calls", "total_ns", "avg_ns", "percent"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])

def pmc(sub, counter):
    path = find(sub, "counter_collection.csv")
    acc = {}
    if not path:
        return acc
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            k = short(r["Kernel_Name"])
            a = acc.setdefault(k, [0.0, 0])
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc

fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
res = {"unit": "bytes per dispatch", "note": "raw counter x 1024; fetch_corrected = 2 x fetch_raw (gfx950 FETCH_SIZE halves wide coalesced reads)", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, [0, 0]); w = write.get(k, [0, 0])
    fr = f[0] / f[1] * 1024 if f[1] else None
    wr = w[0] / w[1] * 1024 if w[1] else None
    res["kernels"][k] = {"dispatches": f[1] or w[1], "fetch_raw": fr, "fetch_corrected": (2 * fr if fr is not None else None), "write": wr}
bench = os.path.join(out, f"{tag}_bench_under_stats.json")
try:
    line = [l for l in open(bench) if l.startswith("{")][-1]
    res["bench_line_under_profiler"] = json.loads(line)
except Exception as e:
    res["bench_line_under_profiler"] = str(e)
json.dump(res, open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))
if rows:
    for r in rows[:12]:
        print(short(r["Name"])[:60], r["Calls"], r["AverageNs"], r["Percentage"])
