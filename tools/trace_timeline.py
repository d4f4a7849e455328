#!/usr/bin/env python3
"""Print one steady-state frame of a rocprofv3 --kernel-trace [--memory-copy-trace] CSV set as a timeline."""
import csv, glob, sys
path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_lighting"
events = []
for f in glob.glob(path + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:34]
        events.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, "q" + r["Queue_Id"]))
for f in glob.glob(path + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        events.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:22], "dma"))
events.sort()
idx = [i for i, e in enumerate(events) if marker in e[2]]
a, b = idx[-4], idx[-3]
t0 = events[a][0]
print(f"frame period {(events[b][0] - t0) / 1000:.1f} us")
for s, e, name, q in events[a:b + 1]:
    print(f"{name:36s} {q:>4s} start {(s - t0) / 1000:8.1f} end {(e - t0) / 1000:8.1f} dur {(e - s) / 1000:7.1f}")
