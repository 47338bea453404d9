#!/bin/bash
# same-box comparison of several builds of libvislam_hip.so, alternating runs: tools/multi_bench.sh REPS LIB_A LIB_B [LIB_C ...]
# prints value and kernels_ms_per_step of every run (bench.py --no-legs --no-cpu-baseline)
REPS=$1; shift
for r in $(seq $REPS); do
  for L in "$@"; do
    VISLAM_HIP_LIB=$L timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --steps 60 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=j['ms']
print('$L'.split('/')[-1].ljust(24), round(j['value']), ' '.join(f'{a}={b:.3f}' for a,b in k.items()))"
  done
done
