#!/usr/bin/env python
"""Average the PMC counters of the candidate kernel over its dispatches (rocprofv3 --pmc csv output)."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "knn_select"
acc = {}
for fn in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    with open(fn) as f:
        for row in csv.DictReader(f):
            if pat not in row["Kernel_Name"]:
                continue
            key = row["Counter_Name"]
            dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            a = acc.setdefault(key, [0.0, 0, 0.0])
            a[0] += float(row["Counter_Value"])
            a[1] += 1
            a[2] += dur
out = {k: {"mean": v[0] / v[1], "dispatches": v[1], "mean_ns": v[2] / v[1]} for k, v in acc.items()}
print(json.dumps(out, indent=1))
with open(os.path.join(root, "summary.json"), "w") as f:
    json.dump(out, f, indent=1)
