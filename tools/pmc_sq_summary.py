#!/usr/bin/env python3
"""Per-kernel means of the SQ counter passes of tools/pmc_sq.sh: tools/pmc_sq_summary.py OUTDIR [kernel-prefix ...]"""
import collections
import csv
import glob
import sys

out = sys.argv[1]
want = sys.argv[2:] or ["k_fast", "k_describe", "k_resize", "k_select", "k_knn_mfma"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for w in want:
            if r["Kernel_Name"].startswith(w):
                acc[w][r["Counter_Name"]].append(float(r["Counter_Value"]))
for w in want:
    print(w, {k: round(sum(v) / len(v)) for k, v in sorted(acc[w].items())}, "calls", {k: len(v) for k, v in acc[w].items()}.get("SQ_WAVE_CYCLES"))
