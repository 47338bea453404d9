#!/bin/bash
# kernel-trace timeline of the headline with the results download queued behind every step: tools/experiments/r5_trace_d2h.sh
export TMPDIR=/tmp
D=$GRAFT_REPO_ROOT/gpurun_out/trace_d2h
rm -rf $D; mkdir -p $D
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/tools/step_time.py 15 12 d2h > $D/run.log 2>&1)
tail -1 $D/run.log
python3 $GRAFT_REPO_ROOT/tools/trace_timeline.py $D 110 > $GRAFT_REPO_ROOT/gpurun_out/timeline_d2h.txt
ls $D/*/ | head; head -5 $D/*/*memory_copy_trace.csv 2>/dev/null | cut -c1-200
find $D -name "*.csv" -delete
