#!/bin/bash
# usage (GPU box, repo root): scripts/collect_valu_counters.sh <tag>
# VALU-side counters of the stand-alone kernels and of the pass launches (1/8 degree), three counter passes.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/valu_$tag; rm -rf $out; mkdir -p $out
args="bench.py --steps 3 --warmup 1 --cpu-sample-div 0 --self-check 0 --d2h 0 --launch pass"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/a -o a --output-format csv -- python3 $args > $out/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES -d $out/b -o b --output-format csv -- python3 $args > $out/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_IFETCH -d $out/c -o c --output-format csv -- python3 $args > $out/c.log 2>&1
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for d in "abc":
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            key = next((k for k in ("pass_a", "pass_b", "tail", "quad_kernel<5, 0>", "quad_kernel<5, 1>", "mesh_kernel", "latlon_fused") if k in n), None)
            if key:
                tot[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in tot.items():
    print(k, {c: "%.4g" % (sum(x) / len(x)) for c, x in sorted(v.items())})
PY
