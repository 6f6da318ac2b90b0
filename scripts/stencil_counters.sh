#!/bin/bash
# usage (GPU box, repo root): scripts/stencil_counters.sh <tag>
# Where do the cycles of the generic stencil kernel (midas_tile_kernel, OGG:687-729) go that its VALU does not fill (~20 %)?  LDS issue and
# wait, bank conflicts, scalar-unit activity: one rocprofv3 --pmc pass per counter group of `bench.py --latlon stencil`; a counter the
# chip does not offer fails its own pass only.  Summary: profiles/<tag>_stencil_cycle_counters_r8.md
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/stencil_pmc_${tag}; rm -rf $out; mkdir -p $out
rocprofv3 -L > $out/avail.txt 2>&1 || rocprofv3 --list-avail > $out/avail.txt 2>&1
args="bench.py --workload r8 --latlon stencil --launch kernels --steps 3 --warmup 1 --cpu-sample-div 0 --self-check 0 --d2h 0 --power-probe 0 --checksum 0"
k=0
for grp in "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" "SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
  k=$((k+1))
  rocprofv3 --pmc $grp -d $out/p$k -o p --output-format csv -- python3 $args > $out/p$k.log 2>&1 || echo "pass $k ($grp) failed: see $out/p$k.log"
done
python3 - <<PY
import csv, glob, collections, re
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::|void |\(.*", "", r["Kernel_Name"])
        if "midas" in n or "tile_latlon" in n:
            tot[(n, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
md = ["# Cycle attribution of the generic stencil kernel, $tag, `bench.py --workload r8 --latlon stencil`", "",
      "rocprofv3 --pmc, one pass per group (scripts/stencil_counters.sh); means per launch; SQ counters summed over 8 XCDs x 32 CUs (x 4 SIMDs).",
      "SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count in 4-cycle quanta per the SQ convention; ratios against SQ_BUSY_CYCLES / SQ_WAVE_CYCLES below.", ""]
for key in sorted(tot):
    m = {c: sum(v) / len(v) for c, v in tot[key].items()}
    md += ["## %s, grid %s (%d launches)" % (key[0], key[1], len(next(iter(tot[key].values())))), "", "| counter | mean per launch | / SQ_WAVE_CYCLES | / SQ_ACTIVE_INST_VALU |", "|---|---|---|---|"]
    wc, va = m.get("SQ_WAVE_CYCLES", 0.0), m.get("SQ_ACTIVE_INST_VALU", 0.0)
    for c in sorted(m):
        md.append("| %s | %.4g | %s | %s |" % (c, m[c], ("%.3f" % (m[c] / wc)) if wc else "", ("%.3f" % (m[c] / va)) if va else ""))
    md.append("")
open("profiles/${tag}_stencil_cycle_counters_r8.md", "w").write("\n".join(md) + "\n")
print("\n".join(md))
PY
grep -i "lds\|salu\|sca" $out/avail.txt | head -40
