#!/bin/bash
# Per-kernel durations (rocprofv3 --kernel-trace, last occurrence of each kernel) of one workload for several builds of the library:
#   tools/kernel_ab.sh <workload.py> <kernel-name-prefix> LIB_A [LIB_B ...]     (run on the GPU box, from the repository root)
W=$1; K=$2; shift 2
export TMPDIR=/tmp
for L in "$@"; do
  D=$GRAFT_REPO_ROOT/gpurun_out/kab_$(basename $L .so)
  rm -rf $D; mkdir -p $D
  (cd /tmp && VISLAM_HIP_LIB=$GRAFT_REPO_ROOT/$L timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/$W > /dev/null 2>&1)
  python3 - "$D" "$K" "$L" <<'PY'
import csv, glob, os, sys
d, k, lib = sys.argv[1:4]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last = {}
for r in rows:
    if r["Kernel_Name"].startswith(k) or k == "*":
        last.setdefault(r["Kernel_Name"][:48], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for n, v in last.items():
    print(os.path.basename(lib).ljust(26), n.ljust(48), "last", [round(x, 1) for x in v[-int(os.environ.get("KAB_LAST", "3")):]])
PY
done
