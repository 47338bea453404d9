#!/bin/bash
# kernel-trace timeline of the GPU main's per-frame sequence (legs.gpu_main_sequence): tools/experiments/r5_trace_main.sh
export TMPDIR=/tmp
D=$GRAFT_REPO_ROOT/gpurun_out/trace_main
rm -rf $D; mkdir -p $D
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/tools/main_sequence_time.py 10 > $D/run.log 2>&1)
tail -1 $D/run.log
python3 $GRAFT_REPO_ROOT/tools/trace_timeline.py $D 130 > $GRAFT_REPO_ROOT/gpurun_out/timeline_main.txt
find $D -name "*.csv" -delete
