set -x
cd $GRAFT_REPO_ROOT
T=${1:-r5e}
timeout -k 10 400 python -m pytest tests/test_detect_gpu.py tests/test_edge_cases_gpu.py tests/test_pipeline_determinism_gpu.py tests/test_configs_gpu.py -x -q -m gpu > gpurun_out/${T}_detect.log 2>&1 && \
bash tools/multi_bench.sh 2 vi-slam_amd/lib/libvislam_hip_r4.so vi-slam_amd/lib/libvislam_hip.so > gpurun_out/${T}_ab.log 2>&1 && \
bash tools/r5_pmc_ab.sh k_fast vi-slam_amd/lib/libvislam_hip.so > gpurun_out/${T}_pmc.log 2>&1
tail -3 gpurun_out/${T}_detect.log; cat gpurun_out/${T}_ab.log; cat gpurun_out/${T}_pmc.log
