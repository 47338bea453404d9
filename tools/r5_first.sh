cd $GRAFT_REPO_ROOT
run() { VISLAM_HIP_LIB=$GRAFT_REPO_ROOT/vi-slam_amd/lib/libvislam_hip_knobs.so timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --steps 12 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=j['kernels_ms_per_step']
print('$1'.ljust(12), round(j['value']), ' '.join(f'{a[3:]}={b:.3f}' for a,b in k.items()))"; }
for r in 1 2 3; do
  VIS_RANSAC_FIRST=16 run first16
  VIS_RANSAC_FIRST=8 run first8
  VIS_RANSAC_FIRST=4 run first4
done
