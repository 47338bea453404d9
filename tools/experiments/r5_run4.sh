set -x
cd $GRAFT_REPO_ROOT
T=${1:-r5o}
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/${T}_tests.log 2>&1 && \
timeout -k 10 200 python tools/stress_detect.py 120 9101 > gpurun_out/${T}_stress_detect.log 2>&1 && \
timeout -k 10 200 python tools/stress_batch.py 120 9102 > gpurun_out/${T}_stress_batch.log 2>&1
tail -3 gpurun_out/${T}_tests.log; tail -3 gpurun_out/${T}_stress_detect.log; tail -3 gpurun_out/${T}_stress_batch.log
