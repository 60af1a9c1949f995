#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter CSVs over the launches of the transport and peel kernels and write profiles/<name>.json.
usage: pmc_summary.py OUT.json EVENTS_PER_STEP "COMMAND" DIR [DIR ...]   (one DIR per separate --pmc pass)
Top level = the transport kernel (as bench.py's pmc_traffic expects); "peel_kernel" = the same counters for the peel kernel."""
import csv, glob, json, os, sys
from collections import defaultdict

out, eps, cmd, dirs = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4:]
res = {"peel_kernel": {}}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = {"transport_kernel": defaultdict(lambda: defaultdict(float)), "peel_kernel": defaultdict(lambda: defaultdict(float))}
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            key = "transport_kernel" if "transport_kernel" in name else ("peel_kernel" if "peel_kernel" in name else None)
            if key:
                per[key][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])      # summed over the XCD rows
        for key, tab in per.items():
            for name, disp in tab.items():
                v = list(disp.values())
                (res if key == "transport_kernel" else res["peel_kernel"])[name] = {"launches": len(v), "mean_per_launch": sum(v) / len(v), "total": sum(v)}
res["events_per_step"] = eps
res["command"] = cmd
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
