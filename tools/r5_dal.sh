cd $GRAFT_REPO_ROOT
VISLAM_HIP_LIB=$GRAFT_REPO_ROOT/vi-slam_amd/lib/libvislam_hip_dal.so timeout -k 10 300 python -m pytest tests/test_detect_gpu.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -2
bash tools/multi_bench.sh 2 vi-slam_amd/lib/libvislam_hip.so vi-slam_amd/lib/libvislam_hip_dal.so 2>&1 | cut -c1-190
bash tools/r5_pmc_ab.sh k_describe vi-slam_amd/lib/libvislam_hip.so vi-slam_amd/lib/libvislam_hip_dal.so 2>&1 | cut -c1-900
