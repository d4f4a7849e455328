#!/usr/bin/env python3
"""Host launch time vs GPU start time per kernel (rocprofv3 --kernel-trace --hip-runtime-trace CSVs, joined on correlation id)."""
import csv, glob, sys
path = sys.argv[1]
api = {}
for f in glob.glob(path + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        api[r["Correlation_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"])
rows = []
for f in glob.glob(path + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = api.get(r["Correlation_Id"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:30]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, a))
rows.sort()
idx = [i for i, r in enumerate(rows) if "k_lighting" in r[2]]
a, b = idx[-5], idx[-3]
t0 = rows[a][0]
for s, e, name, ap in rows[a:b + 1]:
    host = f"host launch at {(ap[0] - t0) / 1000:9.1f}" if ap else "host ?"
    print(f"{name:32s} gpu start {(s - t0) / 1000:8.1f} end {(e - t0) / 1000:8.1f}   {host}")
