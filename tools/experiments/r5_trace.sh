#!/bin/bash
# kernel-trace timeline of the pipelined steady state for several builds: tools/experiments/r5_trace.sh LIB_A [LIB_B ...]
export TMPDIR=/tmp
for L in "$@"; do
  T=$(basename $L .so)
  D=$GRAFT_REPO_ROOT/gpurun_out/trace_$T
  rm -rf $D; mkdir -p $D
  (cd /tmp && VISLAM_HIP_LIB=$GRAFT_REPO_ROOT/$L timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/tools/step_time.py 15 12 > $D/run.log 2>&1)
  tail -1 $D/run.log
  python3 $GRAFT_REPO_ROOT/tools/trace_timeline.py $D 110 > $GRAFT_REPO_ROOT/gpurun_out/timeline_$T.txt
  find $D -name "*.csv" -delete
done
