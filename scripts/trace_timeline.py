#!/usr/bin/env python3
"""Print one steady-state frame of a rocprofv3 --kernel-trace CSV as a timeline (start, duration, gap, queue)."""
import csv, glob, sys
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True) if not path.endswith(".csv") else [path]
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "k_lighting"
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-4], idx[-3]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
print(f"frame period {(int(rows[b]['Start_Timestamp']) - t0) / 1000:.1f} us")
busy = 0
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:34]
    print(f"{name:36s} q{r['Queue_Id']:>2s} start {(s - t0) / 1000:8.1f} dur {(e - s) / 1000:7.1f} gap {(s - prev_end) / 1000:7.1f}")
    prev_end = max(prev_end, e)
