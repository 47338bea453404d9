#!/bin/bash
# Harris inside the streaming k_fast? -- the product against two TIMING builds (keypoint selection wrong on purpose)   (GPU box, repo root)
#   libvislam_hip_noharris.so  make TAG=_noharris EXTRA=-DVIS_TIMING_NOHARRIS lib : k_select as if every candidate came with its Harris response for free
#                              (no 9 x 9 pyramid re-read, no response arithmetic): the MOST k_select can gain
#   libvislam_hip_hfast.so     make TAG=_hfast EXTRA="-DVIS_TIMING_HARRIS_IN_FAST -DVIS_TIMING_NOHARRIS" lib : the same + k_fast computes the response of every candidate
#                              it emits from its own LDS pixel ring, one lane per candidate (the form that needs no extra pass structure)
cd $GRAFT_REPO_ROOT
LIBS="vi-slam_amd/lib/libvislam_hip.so vi-slam_amd/lib/libvislam_hip_noharris.so vi-slam_amd/lib/libvislam_hip_hfast.so"
export VIS_PROFILE_BATCH=512 VIS_PROFILE_STEPS=3
for rep in 1 2; do for K in "k_fast(" k_select; do bash tools/kernel_ab.sh tools/profile_workload.py "$K" $LIBS; done; done
for L in vi-slam_amd/lib/libvislam_hip.so vi-slam_amd/lib/libvislam_hip_hfast.so; do
  T=$(basename $L .so)
  VISLAM_HIP_LIB=$GRAFT_REPO_ROOT/$L bash tools/pmc_kernel.sh k_fast $T 2>&1 | tail -1
  VISLAM_HIP_LIB=$GRAFT_REPO_ROOT/$L bash tools/pmc_kernel.sh k_select $T 2>&1 | tail -1
done
