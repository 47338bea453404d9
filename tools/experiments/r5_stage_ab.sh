#!/bin/bash
# headline bench with different stage sets for several builds: tools/experiments/r5_stage_ab.sh "STAGESETS" LIB_A [LIB_B ...]
cd $GRAFT_REPO_ROOT
SETS=$1; shift
for r in 1 2; do for L in "$@"; do for S in $SETS; do
  VISLAM_HIP_LIB=$GRAFT_REPO_ROOT/$L timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --steps 60 --stages $S 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=j['kernels_ms_per_step']
print('$L'.split('/')[-1].ljust(24), 'stages $S', round(j['value']), ' '.join(f'{a[3:]}={b:.3f}' for a,b in k.items()))"
done; done; done
