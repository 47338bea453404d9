#!/bin/bash
# End-of-round profiling on the GPU box: tools/final_profile.sh OUTDIR COMMIT [SECTION ...]   (OUTDIR under gpurun_out/; default: every section)
#   stats  rocprofv3 --kernel-trace --stats of the default bench command, side legs off (they would mix other configurations into the per-kernel
#          averages), and once more with the side legs on (configs 3 / 5, fixed-1000 RANSAC, parallax, alignment): one stats file for the legs
#   pipe   PMC passes (separate runs, no trace domain besides --kernel-trace) of the headline workload -> pipe/pmc_traffic.json
#   legs   the same passes on configs 3 and 5's OWN workloads -> merged into pmc_traffic.json under legs.c3 / legs.c5 (needs `pipe` of this OUTDIR)
#   grad   PMC passes + stats of the gradient stage
#   pose   a PMC pass of the loaded pose legs + tools/f64_rates -> pose/pmc_pose.json
#   bench  the bench record AFTER the counter files of this very run are in place (compact line + full record)
#   micro  the instruction-rate microbenchmarks
set -e
OUT=${1:-gpurun_out/final}
COMMIT=${2:-unknown}
shift 2 || true
SECTIONS=${@:-stats pipe legs grad pose bench micro}
want() { [[ " $SECTIONS " == *" $1 "* ]]; }
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
PIPE_PASSES=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VALU_MFMA_MOPS_F6F4 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_I8")

if want stats; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-legs --full-out $R/$OUT/bench_under_rocprof_full.json > $R/$OUT/bench_under_rocprof.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_legs -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --full-out $R/$OUT/bench_legs_under_rocprof_full.json > $R/$OUT/bench_legs_under_rocprof.log 2>&1
  echo "section stats done"
fi

if want pipe; then
  export VIS_PROFILE_BATCH=512 VIS_PROFILE_STEPS=3 VIS_PROFILE_WARM=2 VIS_PROFILE_CONFIG=headline VIS_PROFILE_N_DESC=1000
  for c in "${PIPE_PASSES[@]}"; do
    n=$(echo $c | cut -d' ' -f1)
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pipe/pass_$n -- python3 $R/tools/profile_workload.py > $R/$OUT/pipe_$n.log 2>&1
  done
  # the measured VALU issue ceilings of this box, same lease (vi-slam_amd/lib/valu_peak is built by `make -C vi-slam_amd/csrc tools`)
  $R/vi-slam_amd/lib/valu_peak > $R/$OUT/valu_peak.log 2>&1
  python3 $R/tools/pmc_summarize.py $R/$OUT/pipe 512 $COMMIT $R/$OUT/valu_peak.log > $R/$OUT/pmc_summary.log 2>&1
  echo "section pipe done"
fi

if want legs; then
  # configs 3 and 5 on their OWN workload (1920x1080 / 4000 kps / 64 frames; 3840x2160 / 8000 kps / 16 frames per launch): traffic, L2 hit,
  # MFMA busy, instruction counts and LDS activity of every kernel; merged into pmc_traffic.json under legs.c3 / legs.c5
  mkdir -p $R/$OUT/pipe
  [ -f $R/$OUT/pipe/pmc_traffic.json ] || cp $R/profiles/pmc_traffic.json $R/$OUT/pipe/pmc_traffic.json      # (legs alone: merged into the committed headline set)
  for leg in c3:64:4000 c5:16:8000; do
    L=${leg%%:*}; rest=${leg#*:}; LB=${rest%%:*}; ND=${rest#*:}
    export VIS_PROFILE_CONFIG=$L VIS_PROFILE_BATCH=$LB VIS_PROFILE_N_DESC=$ND VIS_PROFILE_STEPS=3 VIS_PROFILE_WARM=2
    for c in "${PIPE_PASSES[@]}"; do
      n=$(echo $c | cut -d' ' -f1)
      rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pipe_$L/pass_$n -- python3 $R/tools/profile_workload.py > $R/$OUT/pipe_${L}_$n.log 2>&1
    done
    PMC_LEG=$L PMC_MERGE_INTO=$R/$OUT/pipe/pmc_traffic.json python3 $R/tools/pmc_summarize.py $R/$OUT/pipe_$L $LB $COMMIT > $R/$OUT/pmc_summary_$L.log 2>&1
    echo "leg $L done"
  done
  unset VIS_PROFILE_CONFIG VIS_PROFILE_N_DESC
  echo "section legs done"
fi

if want grad; then
  export VIS_PROFILE_BATCH=1024 VIS_PROFILE_STEPS=3 VIS_PROFILE_WARM=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    n=$(echo $c | cut -d' ' -f1)
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/grad/pass_$n -- python3 $R/tools/profile_gradient.py > $R/$OUT/grad_$n.log 2>&1
  done
  python3 $R/tools/pmc_summarize.py $R/$OUT/grad 1024 $COMMIT > $R/$OUT/pmc_grad_summary.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/grad_stats -- python3 $R/tools/profile_gradient.py > $R/$OUT/grad_stats.log 2>&1
  echo "section grad done"
fi

if want pose; then
  # the pose kernels under load (fixed-1000 S-752, config 3): instruction counts + standalone durations, and the double-precision issue rates
  for leg in f1000:fixed1000_probe.py c3:config3_probe.py; do
    n=${leg%%:*}; py=${leg#*:}
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/$OUT/pose/$n -- python3 $R/tools/$py > $R/$OUT/pose_$n.log 2>&1
  done
  $R/vi-slam_amd/lib/f64_rates > $R/$OUT/f64_rates.log 2>&1
  python3 $R/tools/pmc_pose_summarize.py $R/$OUT/pose $COMMIT $R/$OUT/f64_rates.log > $R/$OUT/pmc_pose_summary.log 2>&1
  echo "section pose done"
fi

if want bench; then
  # the freshly written pmc_traffic.json / pmc_pose.json go into profiles/ of this box's copy, then the full default bench runs
  # (round 4's record quoted the previous run's counters beside new timings)
  [ -f $R/$OUT/pipe/pmc_traffic.json ] && cp $R/$OUT/pipe/pmc_traffic.json $R/profiles/pmc_traffic.json
  [ -f $R/$OUT/pose/pmc_pose.json ] && cp $R/$OUT/pose/pmc_pose.json $R/profiles/pmc_pose.json
  (cd $R && python3 bench.py --steps 20 --warmup 5 --full-out $R/$OUT/bench_full.json > $R/$OUT/bench_line.json 2> $R/$OUT/bench_full.err)
  echo "section bench done"
fi

if want micro; then
  [ -x $R/vi-slam_amd/lib/dpp_rates ] && $R/vi-slam_amd/lib/dpp_rates > $R/$OUT/dpp_rates.log 2>&1
fi
# the csv dumps are large: keep the stats summaries, the json summaries and the logs
find $R/$OUT -name "*_kernel_trace.csv" -delete; find $R/$OUT -name "*counter_collection.csv" -delete; find $R/$OUT -name "*agent_info.csv" -delete
echo final_profile done
