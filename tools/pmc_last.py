#!/usr/bin/env python3
"""Counters of the LAST dispatch of one kernel (exact name, e.g. k_fast but not k_fast_fix) over the SQ passes of tools/pmc_kernel.sh:
the steady state of a batched stream (the first dispatches run at the base FAST threshold).  tools/pmc_last.py OUTDIR KERNEL"""
import csv
import glob
import sys

out, kern = sys.argv[1], sys.argv[2]
res = {}
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].split("(")[0].replace("void ", "").strip() == kern]
    if not rows:
        continue
    last = max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            res[r["Counter_Name"]] = res.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            res["_us_" + f.split("/")[-3]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
            res["_vgpr"] = int(r["VGPR_Count"]); res["_lds"] = int(r["LDS_Block_Size"])
print(kern, {k: (round(v, 1) if k.startswith("_") else round(v)) for k, v in sorted(res.items())})
