"""Static instruction statistics of the kernels in a hipcc -S --cuda-device-only listing, per basic-block loop.

usage: python scripts/isa_stats.py file.s [kernel-substring]
Prints, per kernel: total instructions, fp64 VALU instructions, quarter-rate ones (rcp/rsq/sqrt), and the largest
backward-branch loops (label, instruction count inside, fp64 count) -- enough to see what the hot loop costs.
"""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
start = [i for i, l in enumerate(txt) if re.match(r"^(_Z|k_)\w+:", l)]
for si, s in enumerate(start):
    name = txt[s].split(":")[0]
    if want not in name:
        continue
    end = next((i for i in range(s, len(txt)) if txt[i].startswith(".Lfunc_end")), len(txt))
    lines = txt[s:end]
    labels = {}
    ins = []  # (index, mnemonic, operand text)
    for l in lines:
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if l.startswith("\t") and not l.strip().startswith((".", ";")):
            parts = l.strip().split(None, 1)
            ins.append((parts[0], parts[1] if len(parts) > 1 else ""))
    c = Counter(m for m, _ in ins)
    f64 = lambda seq: sum(1 for m, _ in seq if "f64" in m)
    quarter = lambda seq: sum(1 for m, _ in seq if m.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")))
    print("%s\n  total %d, f64 %d, rcp/rsq/sqrt %d, v_div_* %d, branches %d" % (
        name, len(ins), f64(ins), quarter(ins), sum(v for k, v in c.items() if k.startswith("v_div_")),
        sum(v for k, v in c.items() if k.startswith("s_cbranch"))))
    loops = []
    for k, (m, ops) in enumerate(ins):
        if m.startswith("s_cbranch") or m == "s_branch":
            tgt = ops.split()[-1]
            if tgt in labels and labels[tgt] <= k:
                seq = ins[labels[tgt]:k + 1]
                loops.append((len(seq), tgt, f64(seq), quarter(seq)))
    for n, tgt, nf, nq in sorted(loops, reverse=True)[:6]:
        print("    loop %-12s %5d instr, f64 %5d, rcp/rsq/sqrt %3d" % (tgt, n, nf, nq))
