#!/bin/bash
# kernel-trace timeline of config 3 (1080p, 4000 keypoints, 2000 fixed RANSAC iterations, 128 frames per step): tools/experiments/r5_trace_c3.sh
export TMPDIR=/tmp
D=$GRAFT_REPO_ROOT/gpurun_out/trace_c3
rm -rf $D; mkdir -p $D
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/tools/config3_probe.py > $D/run.log 2>&1)
tail -1 $D/run.log | cut -c1-300
python3 $GRAFT_REPO_ROOT/tools/trace_timeline.py $D 400 > $GRAFT_REPO_ROOT/gpurun_out/timeline_c3.txt
find $D -name "*.csv" -delete
