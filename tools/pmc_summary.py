#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean counter value per dispatch."""
import csv, glob, os, sys, collections

def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0].replace("void ", "")[:40]

root = sys.argv[1]
table = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per_dispatch = collections.defaultdict(float)
    names = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            key = (path, row["Dispatch_Id"])
            names[key] = short(row["Kernel_Name"])
            per_dispatch[(key, row["Counter_Name"])] += float(row["Counter_Value"])
    for (key, counter), value in per_dispatch.items():
        table[names[key]][counter].append(value)
counters = sorted({c for k in table.values() for c in k})
import json
with open(os.path.join(root, "summary.json"), "w") as f:
    json.dump({k: {c: sum(v) / len(v) for c, v in vals.items()} for k, vals in table.items()}, f, indent=1)
print("kernel".ljust(42) + "".join(c[-18:].rjust(20) for c in counters))
for kernel, vals in sorted(table.items()):
    line = kernel.ljust(42)
    for c in counters:
        v = vals.get(c)
        line += (f"{sum(v) / len(v):.4g}" if v else "-").rjust(20)
    print(line)
