"""Print the kernel timeline of the last few steps from a rocprofv3 --kernel-trace CSV (start offsets in microseconds).

usage: python scripts/trace_window.py <..._kernel_trace.csv> [n_kernels]
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    name = name[name.find("::") + 2 if "::" in name else 0:][:60]
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>3}  {name}")
