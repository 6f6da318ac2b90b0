"""Instruction mix of the largest loop of one kernel in a hipcc -S --cuda-device-only listing.
usage: python scripts/isa_loop_hist.py file.s kernel-substring"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
s = next(i for i, l in enumerate(txt) if re.match(r"^_Z\w*%s\w*:" % re.escape(sys.argv[2]), l))
e = next(i for i in range(s, len(txt)) if txt[i].startswith(".Lfunc_end"))
body = txt[s:e]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\w+):", l)] if m}
best = (0, 0, 0)
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\w+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] <= i and i - labels[m.group(1)] > best[0]:
        best = (i - labels[m.group(1)], labels[m.group(1)], i)
loop = body[best[1]:best[2] + 1]
c = collections.Counter(l.split()[0] for l in loop if l.startswith("\t") and not l.strip().startswith((".", ";")))
cat = collections.Counter()
for k, v in c.items():
    if k.startswith("s_"):
        cat["scalar " + ("s_mov" if k.startswith("s_mov") else "s_load" if k.startswith("s_load") else "s_waitcnt" if k.startswith("s_wait") else
                         "branch" if "branch" in k else "s_nop" if k == "s_nop" else "other")] += v
    elif k.startswith("v_"):
        cat["VALU " + ("f64" if "f64" in k else "mov" if k.startswith("v_mov") else "cndmask" if "cndmask" in k else "cmp" if k.startswith("v_cmp") else
                       "lane" if "lane" in k else "other")] += v
    else:
        cat["mem " + k] += v
print("largest loop: %d instructions, VALU %d" % (sum(c.values()), sum(v for k, v in c.items() if k.startswith("v_"))))
for k, v in sorted(cat.items(), key=lambda x: -x[1]):
    print("  %-28s %5d" % (k, v))
print("  " + ", ".join("%s %d" % kv for kv in c.most_common(30)))
