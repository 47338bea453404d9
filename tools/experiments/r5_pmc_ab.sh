#!/bin/bash
# SQ counter passes of one kernel for several library builds: tools/experiments/r5_pmc_ab.sh KERNEL LIB_A [LIB_B ...]   (GPU box, repo root)
K=$1; shift
for L in "$@"; do
  T=$(basename $L .so)
  VISLAM_HIP_LIB=$GRAFT_REPO_ROOT/$L bash $GRAFT_REPO_ROOT/tools/pmc_kernel.sh $K $T 2>&1 | tail -1
done
