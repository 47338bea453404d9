cd $GRAFT_REPO_ROOT
S=${1:-7001}
for t in detect batch match align pose; do
  timeout -k 10 400 python tools/stress_$t.py ${2:-150} $((S++)) > gpurun_out/r5u_stress_$t.log 2>&1
  tail -1 gpurun_out/r5u_stress_$t.log
done
