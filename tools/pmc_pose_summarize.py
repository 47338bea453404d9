#!/usr/bin/env python3
"""tools/pmc_pose_summarize.py OUTDIR COMMIT [F64_RATES_TXT] -- per pose kernel of the loaded legs (fixed-1000 S-752, config 3): vector /
scalar wave-instructions per launch (SQ_INSTS_VALU / SQ_INSTS_SALU) and the launch duration of the SAME run (a counter pass serialises
the kernels: standalone durations), from the rocprofv3 --pmc --kernel-trace outputs under OUTDIR/{f1000,c3}; the double-precision issue
ceiling from tools/f64_rates.hip's output (Horner pair mul + add, 2 waves per SIMD, 4 chains) when given.  Writes OUTDIR/pmc_pose.json."""
import collections
import csv
import glob
import json
import os
import sys

out_dir, commit = sys.argv[1], sys.argv[2]
rates = sys.argv[3] if len(sys.argv) > 3 else None
res = {"measured_at_commit": commit, "legs": {}}
KERNELS = ("k_ransac_hyp_list", "k_ransac_hyp", "k_hyp_roots_packed", "k_hyp_roots", "k_hyp_models", "k_hyp_score", "k_ransac_scan", "k_pose_svd",
           "k_pose_final", "k_pose_prep")
for leg in ("f1000", "c3"):
    d = os.path.join(out_dir, leg)
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if (r["Dispatch_Id"]) not in seen:
                seen.add(r["Dispatch_Id"])
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    kernels = {}
    for k in KERNELS:
        if k not in cnt:
            continue
        # a leg launches a kernel once per step for the work list (and once for the first chunk in adaptive runs): keep the loaded launches,
        # i.e. those within a factor 4 of the longest
        longest = max(dur[k])
        keep = [i for i, x in enumerate(dur[k]) if x * 4 >= longest]
        ms = sum(dur[k][i] for i in keep) / len(keep)
        ent = {"launches": len(keep), "ms": ms}
        for c, v in cnt[k].items():
            vv = [v[i] for i in keep if i < len(v)]
            ent[c] = sum(vv) / max(len(vv), 1)
        if "SQ_INSTS_VALU" in ent and ms > 0:
            ent["valu_wave_insts_per_s"] = ent["SQ_INSTS_VALU"] / (ms * 1e-3)
        kernels[k] = ent
    res["legs"][leg] = kernels
if rates and os.path.exists(rates):
    best = {}
    for line in open(rates):
        if "wave-instr/s" not in line:
            continue
        name = line.split("chains=")[0].strip()
        ch = int(line.split("chains=")[1].split()[0]); w = int(line.split("w/simd=")[1].split()[0])
        v = float(line.split("ms")[1].split("wave-instr/s")[0])
        best[(name, ch, w)] = v
    pick = {n: v for (n, ch, w), v in best.items() if ch == 4 and w == 2}
    res["f64_issue_ceiling"] = {"unit": "wave-instr/s", "by_instruction_2_waves_per_simd_4_chains": pick,
                                "by_instruction_8_waves_per_simd_8_chains": {n: v for (n, ch, w), v in best.items() if ch == 8 and w == 8},
                                "what": "tools/f64_rates.hip in the same lease"}
json.dump(res, open(os.path.join(out_dir, "pmc_pose.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
