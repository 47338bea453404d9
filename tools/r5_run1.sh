set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_detect_gpu.py -x -q -m gpu > gpurun_out/r5a_detect.log 2>&1 && \
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r5a_tests.log 2>&1 && \
bash tools/multi_bench.sh 2 vi-slam_amd/lib/libvislam_hip_r4.so vi-slam_amd/lib/libvislam_hip.so > gpurun_out/r5a_ab.log 2>&1
tail -5 gpurun_out/r5a_detect.log; tail -5 gpurun_out/r5a_tests.log; cat gpurun_out/r5a_ab.log
