#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter CSVs over the transport-kernel launches and write profiles/<name>.json.
usage: pmc_summary.py OUT.json EVENTS_PER_STEP "COMMAND" DIR [DIR ...]   (one DIR per separate --pmc pass)"""
import csv, glob, json, os, sys
from collections import defaultdict

out, eps, cmd, dirs = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4:]
res = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(lambda: defaultdict(float))                 # counter -> dispatch -> value (summed over XCD rows)
        for row in csv.DictReader(open(f)):
            if "transport_kernel" not in row.get("Kernel_Name", ""):
                continue
            per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for name, disp in per.items():
            v = list(disp.values())
            res[name] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
res["events_per_step"] = eps
res["command"] = cmd
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
