#!/bin/bash
# one SQ counter pass over the pipeline workload, one kernel only: tools/pmc_kernel.sh KERNEL TAG
KERNEL=${1:-k_fast}
TAG=${2:-x}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_${KERNEL}_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VIS_PROFILE_BATCH=512 VIS_PROFILE_STEPS=2
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/p1 -- python3 $R/tools/profile_workload.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/p3 -- python3 $R/tools/profile_workload.py > $OUT/p3.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p2 -- python3 $R/tools/profile_workload.py > $OUT/p2.log 2>&1
python3 $R/tools/pmc_last.py $OUT $KERNEL
