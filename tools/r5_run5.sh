set -x
cd $GRAFT_REPO_ROOT
T=${1:-r5p}
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/${T}_tests.log 2>&1
tail -16 gpurun_out/${T}_tests.log
