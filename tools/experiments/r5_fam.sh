# k_fast alone (detect stage only) for several builds: tools/experiments/r5_fam.sh LIB...
cd $GRAFT_REPO_ROOT
for L in "$@"; do
  VISLAM_HIP_LIB=$GRAFT_REPO_ROOT/$L timeout -k 10 200 python bench.py --no-legs --no-cpu-baseline --steps 6 --stages 1 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=j['kernels_ms_per_step']
print('$L'.split('/')[-1].ljust(24), round(j['value']), 'fast=%.3f' % k['ms_fast'])"
done
