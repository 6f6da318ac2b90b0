#!/bin/bash
# usage: scripts/pass_trace.sh <tag> [bench args...]   (environment knobs are inherited)
# kernel trace of the fused pass; prints the median duration of each pass kernel
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/pt_$tag -o t --output-format csv -- python3 bench.py --steps 30 --warmup 3 --cpu-sample-div 0 --launch pass --graph 0 "$@" > gpurun_out/pt_$tag.log 2>&1
python3 - <<PY
import csv, statistics
rows = list(csv.DictReader(open("gpurun_out/pt_$tag/t_kernel_trace.csv")))
d = {}
for r in rows:
    n = r["Kernel_Name"]
    if "pass_" in n or "tail" in n:
        k = "A" if "pass_a" in n else ("B" if "pass_b" in n else "C")
        d.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("$tag", {k: round(statistics.median(v[:12]), 1) for k, v in sorted(d.items())}, "first 12 launches")
PY
