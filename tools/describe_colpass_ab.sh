#!/bin/bash
# k_describe: the product against two TIMING builds (their descriptors are wrong on purpose) -- tools/describe_colpass_ab.sh   (GPU box, repo root)
#   libvislam_hip_oneread.so  make TAG=_oneread EXTRA=-DVIS_TIMING_ONEREAD lib : one 16-bit gather per sample, no vertical taps (sample phase of a full 2D blur, column pass FREE)
#   libvislam_hip_colpass.so  make TAG=_colpass EXTRA=-DVIS_TIMING_COLPASS lib : the same sample phase (one byte gather) WITH the column pass's cost emulated: planes from the
#                             row pass's accumulators (2 v_perm + 2 v_xor per tile), 18 more v_mfma_i32_16x16x64_i8, combine / round / saturate / pack, 9 dword stores; no u16 buffer
# Kernel durations (kernel trace, isolated launches) and the SQ counters of the last dispatch, per build.
cd $GRAFT_REPO_ROOT
LIBS="vi-slam_amd/lib/libvislam_hip.so vi-slam_amd/lib/libvislam_hip_oneread.so vi-slam_amd/lib/libvislam_hip_colpass.so"
export VIS_PROFILE_BATCH=512 VIS_PROFILE_STEPS=3
for rep in 1 2; do bash tools/kernel_ab.sh tools/profile_workload.py k_describe $LIBS; done
for L in $LIBS; do
  T=$(basename $L .so)
  VISLAM_HIP_LIB=$GRAFT_REPO_ROOT/$L bash tools/pmc_kernel.sh k_describe $T 2>&1 | tail -1
done
