#!/bin/bash
# SQ-side PMC passes for the detect kernels (run on the GPU box): tools/pmc_sq.sh OUTDIR
set -e
OUT=${1:-gpurun_out/pmcsq}
mkdir -p $OUT
export VIS_PROFILE_BATCH=512 VIS_PROFILE_STEPS=2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/$OUT/avail.txt 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/$OUT/p1 -- python3 $R/tools/profile_workload.py > $R/$OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $R/$OUT/p2 -- python3 $R/tools/profile_workload.py > $R/$OUT/p2.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $R/$OUT/p3 -- python3 $R/tools/profile_workload.py > $R/$OUT/p3.log 2>&1
