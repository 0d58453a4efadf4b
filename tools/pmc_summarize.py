#!/usr/bin/env python
"""Condense the rocprofv3 --pmc passes written by tools/pmc_run.sh into one JSON:
python tools/pmc_summarize.py mode=dir [mode=dir ...] > profiles/rNN/pmc_summary.json
For every kernel the counters of its FIRST dispatch in each pass (a full 32,768-chunk decoder launch for the bench
workload); Grid_Size / Workgroup_Size / Scratch_Size / VGPR_Count are kept so the launch can be identified."""
import csv, glob, json, os, sys

out = {}
for arg in sys.argv[1:]:
    mode, d = arg.split("=", 1)
    per = {}
    for path in sorted(glob.glob(os.path.join(d, "pmc*_counter_collection.csv"))):
        seen = {}
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0][:110]
            first = seen.setdefault(name, row["Dispatch_Id"])
            if row["Dispatch_Id"] != first:
                continue
            k = per.setdefault(name, {})
            k[row["Counter_Name"]] = float(row["Counter_Value"])
            k.setdefault("_launch", {"grid": int(row["Grid_Size"]), "workgroup": int(row["Workgroup_Size"]),
                                     "scratch_bytes_per_lane": int(row["Scratch_Size"]), "vgprs": int(row["VGPR_Count"]),
                                     "lds_bytes": int(row["LDS_Block_Size"])})
    out[mode] = per
json.dump(out, sys.stdout, indent=1)
