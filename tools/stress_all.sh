#!/bin/bash
# every randomised parity tool for SECONDS each with consecutive seeds from SEED0, then two soak runs: tools/stress_all.sh OUTDIR [SEED0 [SECONDS]]   (GPU box)
cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/stress}; S=${2:-7001}; T=${3:-120}
mkdir -p $OUT
for t in detect batch match align pose; do
  timeout -k 10 $((T + 250)) python tools/stress_$t.py $T $((S++)) > $OUT/stress_$t.log 2>&1
  tail -1 $OUT/stress_$t.log
done
for mode in parallax main; do
  timeout -k 10 $((T + 250)) python tools/soak_pipeline.py $((T / 2)) 256 12 $mode > $OUT/soak_$mode.log 2>&1
  tail -1 $OUT/soak_$mode.log
done
