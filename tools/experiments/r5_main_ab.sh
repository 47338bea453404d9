#!/bin/bash
# the GPU main's per-frame sequence (legs.gpu_main_sequence) and the headline for several builds, alternating: tools/experiments/r5_main_ab.sh LIB_A LIB_B ...
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for L in "$@"; do
  echo -n "$(basename $L) "; VISLAM_HIP_LIB=$GRAFT_REPO_ROOT/$L timeout -k 10 200 python tools/main_sequence_time.py 40 2>/dev/null | tail -1
done; done
bash tools/multi_bench.sh 2 "$@" | cut -c1-150
