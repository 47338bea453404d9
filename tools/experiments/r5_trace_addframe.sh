#!/bin/bash
# kernel + copy timeline of VISystemGPU::AddFrameGPU on the adapter classes (vi-slam_amd/lib/addframe_bench): tools/experiments/r5_trace_addframe.sh
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
D=$GRAFT_REPO_ROOT/gpurun_out/trace_addframe
rm -rf $D; mkdir -p $D
python - <<'PY'
import os
src = open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "bench.py")).read()
open("/tmp/cal.xml", "w").write(src.split('CAL_XML = """', 1)[1].split('"""', 1)[0])
PY
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $D -- $GRAFT_REPO_ROOT/vi-slam_amd/lib/addframe_bench /tmp/cal.xml 60 6 > $D/run.log 2>&1)
grep "^{" $D/run.log | cut -c1-200
python3 - <<'PY'
import csv, glob, os
D = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/trace_addframe")
ev = []
for f in glob.glob(D + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
for f in glob.glob(D + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"]))
ev.sort()
# the last 20 frames: a frame starts at a k_half... find frame boundaries by the first kernel of Camera::Update (k_half_all / k_half4)
tail = ev[-2200:]
t0 = tail[0][0]
busy = sum(e - s for s, e, _ in tail)
span = tail[-1][1] - t0
print("last %d device operations: span %.2f ms, device busy %.2f ms (%.0f %%)" % (len(tail), span / 1e6, busy / 1e6, 100.0 * busy / span))
out = open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/timeline_addframe.txt"), "w")
for s, e, n in tail[-120:]: out.write("%10.1f %8.1f %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
find $D -name "*.csv" -delete
