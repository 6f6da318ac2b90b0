#!/bin/bash
# usage (GPU box, repo root): scripts/valu_counters.sh <tag> <workload> [extra bench.py arguments]
# VALU-side counters of every kernel of a bench run (the pass launches and the stand-alone kernels of its last, per-kernel pass),
# three counter passes; prints one line per kernel: mean counter values per launch.
tag=$1; wl=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/valu_${tag}_${wl}; rm -rf $out; mkdir -p $out
args="bench.py --workload $wl --steps 3 --warmup 1 --cpu-sample-div 0 --self-check 0 --d2h 0 --power-probe 0 --dp-arc-other 0 --launch pass --tune-strips 0 $*"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/a -o a --output-format csv -- python3 $args > $out/a.log 2>&1 &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES -d $out/b -o b --output-format csv -- python3 $args > $out/b.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR -d $out/c -o c --output-format csv -- python3 $args > $out/c.log 2>&1
python3 - <<PY > $out/summary.txt
import csv, glob, collections, re, json, os
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for d in "abc":
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(anonymous namespace\)::|void |\(.*", "", r["Kernel_Name"])
            tot[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
rec = {}
md = ["# VALU counters per launch, $tag, workload $wl $*", "",
      "rocprofv3 --pmc, three passes (scripts/valu_counters.sh).  SQ counters are summed over the 8 XCDs x 32 CUs x 4 SIMDs; GRBM_GUI_ACTIVE over the",
      "8 XCDs.  SQ_ACTIVE_INST_VALU counts 4-cycle issue quanta (a quarter-rate fp64 rcp/rsq/sqrt counts 4), so",
      "VALU busy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8).  Means over the launches of the profiled command.", "",
      "| kernel | launches | wave64 VALU instr | of which fp64 transcendental | VALU busy | GUI cycles / XCD | waves | SALU instr |", "|---|---|---|---|---|---|---|---|"]
for k, v in sorted(tot.items()):
    m = {c: sum(x) / len(x) for c, x in v.items()}
    if m.get("SQ_INSTS_VALU", 0) < 1e5:
        continue
    busy = m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (m["GRBM_GUI_ACTIVE"] / 8) if m.get("GRBM_GUI_ACTIVE") else 0
    rec[k] = {"launches": len(v["SQ_INSTS_VALU"]), "wave64_valu_instr": m["SQ_INSTS_VALU"], "fp64_transcendental_instr": m.get("SQ_INSTS_VALU_TRANS_F64", 0),
              "valu_busy_frac": busy, "gui_cycles_per_xcd": m.get("GRBM_GUI_ACTIVE", 0) / 8, "waves": m.get("SQ_WAVES", 0), "salu_instr": m.get("SQ_INSTS_SALU", 0),
              "wave_cycles_quads": m.get("SQ_WAVE_CYCLES", 0), "wait_any_quads": m.get("SQ_WAIT_ANY", 0), "wait_inst_any_quads": m.get("SQ_WAIT_INST_ANY", 0)}
    md.append("| \`%s\` | %d | %.4g | %.4g | %.1f %% | %.4g | %d | %.4g |" % (k[:60], rec[k]["launches"], m["SQ_INSTS_VALU"], m.get("SQ_INSTS_VALU_TRANS_F64", 0),
              100 * busy, m.get("GRBM_GUI_ACTIVE", 0) / 8, m.get("SQ_WAVES", 0), m.get("SQ_INSTS_SALU", 0)))
    print("%-48s launches=%d valu=%.4g trans64=%.4g busy=%.1f%% gui/xcd=%.4g waves=%d wavecyc=%.4g wait_any=%.4g wait_inst=%.4g salu=%.4g" % (
        k[:48], len(v["SQ_INSTS_VALU"]), m["SQ_INSTS_VALU"], m.get("SQ_INSTS_VALU_TRANS_F64", 0), 100 * busy, m.get("GRBM_GUI_ACTIVE", 0) / 8,
        m.get("SQ_WAVES", 0), m.get("SQ_WAVE_CYCLES", 0), m.get("SQ_WAIT_ANY", 0), m.get("SQ_WAIT_INST_ANY", 0), m.get("SQ_INSTS_SALU", 0)))
key = "$wl" + ("" if not "$*".strip() else " " + "$*".strip())
jf = "profiles/valu_counters.json"
allj = json.load(open(jf)) if os.path.exists(jf) else {}
import ctypes
_lib = ctypes.CDLL("ocean_model_grid_generator_amd/csrc/libogg_hip.so"); _lib.ogg_version.restype = ctypes.c_char_p
rec["_lib_src_hash"] = _lib.ogg_version().decode().split(" src ")[-1]
allj[key] = rec
allj["_source"] = "scripts/valu_counters.sh (rocprofv3 --pmc, a builder-side run of bench.py, NOT the process that printed a bench line); last updated for $tag"
json.dump(allj, open(jf, "w"), indent=1, sort_keys=True)
open("profiles/${tag}_valu_counters_${wl}.md", "w").write("\n".join(md) + "\n")
PY
cat $out/summary.txt
