"""Condense the three counter passes of scripts/collect_valu_counters.sh into profiles/<tag>_valu_counters_r8.md.

usage: python scripts/summarize_valu.py <tag>      (reads gpurun_out/valu_<tag>/)
"""
import collections
import csv
import glob
import sys

tag = sys.argv[1]
out = "gpurun_out/valu_%s" % tag
tot = collections.defaultdict(lambda: collections.defaultdict(list))
KEYS = ("pass_a", "pass_b", "tail", "quad_kernel<5, 0>", "quad_kernel<5, 1>", "mesh_kernel", "latlon_fused")
for d in "abc":
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, d), recursive=True):
        for r in csv.DictReader(open(f)):
            key = next((k for k in KEYS if k in r["Kernel_Name"]), None)
            if key:
                tot[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = {"pass_a": "pass_a_kernel<5>", "pass_b": "pass_b_kernel<5>", "tail": "bipolar_quad_tail_kernel<5>",
         "quad_kernel<5, 0>": "bipolar_quad_kernel<5,FAST>", "quad_kernel<5, 1>": "bipolar_quad_kernel<5,GUARD>",
         "mesh_kernel": "bipolar_mesh_kernel<false>", "latlon_fused": "latlon_fused_kernel"}
lines = ["# VALU counters per launch, %s, workload r8 (1/8 degree, 1 GPU)" % tag, "",
         "rocprofv3 --pmc, three passes (scripts/collect_valu_counters.sh).  SQ counters are summed over the 8 XCDs x 32 CUs x 4 SIMDs;",
         "GRBM_GUI_ACTIVE is summed over the 8 XCDs.  SQ_ACTIVE_INST_VALU counts 4-cycle issue quanta (a quarter-rate fp64 rcp/rsq/sqrt",
         "counts 4), so VALU busy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8).", "",
         "| kernel | wave64 VALU instr | of which fp64 transcendental | VALU issue quanta | GUI cycles / XCD | VALU busy | waves | SALU instr |",
         "|---|---|---|---|---|---|---|---|"]
for k in ("pass_b", "quad_kernel<5, 0>", "quad_kernel<5, 1>", "mesh_kernel", "tail", "latlon_fused", "pass_a"):
    v = {c: sum(x) / len(x) for c, x in tot[k].items()}
    busy = v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (v["GRBM_GUI_ACTIVE"] / 8)
    lines.append("| `%s` | %.4g | %.4g | %.4g | %.4g | %.1f %% | %d | %.4g |" % (
        names[k], v["SQ_INSTS_VALU"], v["SQ_INSTS_VALU_TRANS_F64"], v["SQ_ACTIVE_INST_VALU"], v["GRBM_GUI_ACTIVE"] / 8, 100 * busy,
        v["SQ_WAVES"], v["SQ_INSTS_SALU"]))
open("profiles/%s_valu_counters_r8.md" % tag, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
