#!/bin/bash
# HBM read / write counters of one kernel (last dispatch) for several builds: tools/pmc_fetch.sh KERNEL LIB...   (GPU box, repo root)
K=$1; shift
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp VIS_PROFILE_BATCH=512 VIS_PROFILE_STEPS=2
for L in "$@"; do
  T=$(basename $L .so); OUT=$R/gpurun_out/pmcf_${K}_$T; rm -rf $OUT; mkdir -p $OUT
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && VISLAM_HIP_LIB=$R/$L rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 $R/tools/profile_workload.py > $OUT/$c.log 2>&1)
  done
  echo -n "$T "; python3 $R/tools/pmc_last.py $OUT $K
  find $OUT -name "*.csv" -delete
done
