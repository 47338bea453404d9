set -e
R=$GRAFT_REPO_ROOT; OUT=gpurun_out/final_r4c; COMMIT=$1
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
for leg in f1000:fixed1000_probe.py c3:config3_probe.py; do
  n=${leg%%:*}; py=${leg#*:}
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/$OUT/pose/$n -- python3 $R/tools/$py > $R/$OUT/pose_$n.log 2>&1
done
$R/vi-slam_amd/lib/f64_rates > $R/$OUT/f64_rates.log 2>&1
python3 $R/tools/pmc_pose_summarize.py $R/$OUT/pose $COMMIT $R/$OUT/f64_rates.log > $R/$OUT/pmc_pose_summary.log 2>&1
tail -5 $R/$OUT/pmc_pose_summary.log
