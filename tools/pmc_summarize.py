#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one directory per counter pass) into profiles/pmc_traffic.json.
usage: pmc_summarize.py OUTDIR B [COMMIT [VALU_PEAK_LOG]]      where OUTDIR holds pass_*/**/*counter_collection.csv
Environment: VIS_PROFILE_WARM / VIS_PROFILE_STEPS as the workload ran; VIS_PROFILE_N_DESC = descriptors per frame of the workload (1000);
PMC_LEG=<name> + PMC_MERGE_INTO=<pmc_traffic.json>: the set belongs to another configuration (c3, c5: tools/profile_workload.py with
VIS_PROFILE_CONFIG) and is stored under legs.<name> of that file instead of replacing its top level (bench.py quotes a leg's counters only
from the set taken on that leg's own workload)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir, B = sys.argv[1], int(sys.argv[2])
commit = sys.argv[3] if len(sys.argv) > 3 else None      # the build the counters were taken on (stamp it)
peak_log = sys.argv[4] if len(sys.argv) > 4 else None    # stdout of lib/valu_peak run in the same lease (its last line is JSON)
# the workload runs WARM untimed steps before its measured ones (tools/profile_workload.py: the FAST threshold prediction starts
# with the second batch of a stream): the first warm / (warm + steps) of every kernel's dispatches are dropped
warm = int(os.environ.get("VIS_PROFILE_WARM", "2"))
steps = int(os.environ.get("VIS_PROFILE_STEPS", "3"))
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]      # "void k_resize<true>(..." -> "k_resize"
        acc[name][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
summary = {}
for k, cs in acc.items():
    summary[k] = {}
    for c, v in cs.items():
        vals = [x for _, x in sorted(v)]
        if k.startswith("k_") and len(vals) % (warm + steps) == 0 and len(vals) >= warm + steps:
            vals = vals[len(vals) // (warm + steps) * warm:]
        summary[k][c] = {"calls": len(vals), "mean": sum(vals) / len(vals)}
# calibration: the 256 MiB device-to-device copy = the largest copyBuffer dispatch of each pass
known = 256 << 20
cal = {}
for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "copyBuffer" in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            cal[r["Counter_Name"]] = max(cal.get(r["Counter_Name"], 0.0), float(r["Counter_Value"]))
res = {"batch_frames": B, "n_desc": int(os.environ.get("VIS_PROFILE_N_DESC", "1000")), "workload": os.environ.get("VIS_PROFILE_CONFIG", "headline"),
       "steps_counted": steps, "measured_at_commit": commit, "raw": summary,
       "calibration": {"known_bytes_each_way": known, "FETCH_SIZE_units": cal.get("FETCH_SIZE"), "WRITE_SIZE_units": cal.get("WRITE_SIZE"),
                       "fetch_bytes_per_unit": known / cal["FETCH_SIZE"] if cal.get("FETCH_SIZE") else None,
                       "write_bytes_per_unit": known / cal["WRITE_SIZE"] if cal.get("WRITE_SIZE") else None,
                       "note": "MI355X_MICROARCH.md 'HBM': on gfx950 FETCH_SIZE (KB) reports half of a coalesced streaming read, "
                               "WRITE_SIZE (KB) is exact; the factors here are measured on a 256 MiB copy in the same run"}}
if peak_log and os.path.exists(peak_log):
    for line in open(peak_log):
        if line.startswith("{"):
            res.update(json.loads(line))
fb = res["calibration"]["fetch_bytes_per_unit"] or 2048.0
wb = res["calibration"]["write_bytes_per_unit"] or 1024.0
for k in ("k_fast", "k_resize", "k_describe", "k_knn2", "k_knn_mfma", "k_expand", "k_select", "k_select_1024", "k_filter", "k_ransac_hyp", "k_hyp_roots", "k_hyp_roots_packed", "k_hyp_models", "k_hyp_score",
          "k_pose_final", "k_gradient", "k_half4"):
    if k in summary and "FETCH_SIZE" in summary[k]:
        f = summary[k]["FETCH_SIZE"]["mean"]
        w = summary[k].get("WRITE_SIZE", {}).get("mean", 0.0)
        res[k] = {"hbm_bytes_per_launch": f * fb + w * wb, "fetch_units": f, "write_units": w}
        sq = summary[k]
        if "SQ_INSTS_VALU" in sq:
            res[k]["valu_wave_insts_per_launch"] = sq["SQ_INSTS_VALU"]["mean"]
            res[k]["lds_wave_insts_per_launch"] = sq.get("SQ_INSTS_LDS", {}).get("mean")
            res[k]["lds_bank_conflict_cycles"] = sq.get("SQ_LDS_BANK_CONFLICT", {}).get("mean")
            res[k]["salu_wave_insts_per_launch"] = sq.get("SQ_INSTS_SALU", {}).get("mean")
            res[k]["launches_counted"] = sq["SQ_INSTS_VALU"]["calls"]
        if "SQ_LDS_IDX_ACTIVE" in sq and "SQ_BUSY_CYCLES" in sq:
            # SQ_LDS_IDX_ACTIVE: LDS-array cycles summed over the 256 CUs; SQ_BUSY_CYCLES: kernel cycles summed over the 32 shader engines
            # (8 CUs each) -> the fraction of the launch during which a CU's LDS array is busy
            res[k]["lds_idx_active_cycles"] = sq["SQ_LDS_IDX_ACTIVE"]["mean"]
            res[k]["lds_busy_frac"] = sq["SQ_LDS_IDX_ACTIVE"]["mean"] / (sq["SQ_BUSY_CYCLES"]["mean"] * 8.0)
# the whole step: every kernel of the pipeline (k_synth generates the stream, the rocclr kernels are the profiler workload's own copies / fills)
step_valu = step_salu = 0.0
per_kernel = {}
for k, cs in summary.items():
    if not k.startswith("k_") or k in ("k_synth", "k_resize_tab") or "SQ_INSTS_VALU" not in cs:
        continue
    per_step = cs["SQ_INSTS_VALU"]["mean"] * cs["SQ_INSTS_VALU"]["calls"] / steps
    step_valu += per_step
    step_salu += cs.get("SQ_INSTS_SALU", {}).get("mean", 0.0) * cs.get("SQ_INSTS_SALU", {}).get("calls", 0) / steps
    per_kernel[k] = round(per_step)
res["step"] = {"what": "sum over every kernel of one pipelined step (all launches of the step), counter means x launches per step",
               "valu_wave_insts_per_step": step_valu, "salu_wave_insts_per_step": step_salu, "valu_wave_insts_per_kernel": per_kernel, "steps_counted": steps}
leg, target = os.environ.get("PMC_LEG"), os.environ.get("PMC_MERGE_INTO")
if leg and target:
    tj = json.load(open(target))
    tj.setdefault("legs", {})[leg] = {k: v for k, v in res.items() if k not in ("valu_peak_measured",)}
    json.dump(tj, open(target, "w"), indent=1)
    json.dump(res, open(os.path.join(out_dir, f"pmc_leg_{leg}.json"), "w"), indent=1)
else:
    json.dump(res, open(os.path.join(out_dir, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "raw"}, indent=1))
