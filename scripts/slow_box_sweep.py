"""If this box belongs to the slow write class (stand-alone lat-lon kernel below 5.1 TB/s at 1/16 degree: profiles/r04_box_probe.md),
sweep the number of resident lat-lon workgroups of the fused 1/16 degree pass and the helpers; otherwise print the class and stop.
usage: python3 scripts/slow_box_sweep.py [out.jsonl]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "box_probe.py")], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
d = json.loads(p.stdout.strip().splitlines()[-1])
tiles = d["latlon_tiles_r16"]["TBps"]
out = {"tiles_TBps": tiles, "rows_TBps": d["latlon_rows_r16"]["TBps"], "pass_r16": d["fused_pass_r16"], "pass_r8": d["fused_pass_r8"]}
print(out, flush=True)
if tiles < 5.1:
    for extra in ([], ["--set", "OGG_PASS_LL_HELPERS=0"], ["--set", "OGG_PASS_LL_HELPERS=4"]):
        q = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "env_sweep.py"), "--workload", "r16", "--var", "OGG_PASS_LL_WG", "--values",
                            "75", "90", "120", "150", "180", "240", "90"] + extra, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        lines = [l for l in q.stdout.splitlines() if " ms" in l]
        out["sweep " + " ".join(extra)] = lines
        print("\n".join(lines), flush=True)
if len(sys.argv) > 1:
    with open(sys.argv[1], "a") as f:
        f.write(json.dumps(out) + "\n")
