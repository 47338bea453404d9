#!/bin/bash
# A/B of two builds of libvislam_hip.so on ONE box, alternating runs: tools/ab_bench.sh LIB_A LIB_B [REPS]
# prints value and kernels_ms_per_step of every run (bench.py --no-legs --no-cpu-baseline)
A=$1; B=$2; REPS=${3:-3}
for r in $(seq $REPS); do
  for L in $A $B; do
    VISLAM_HIP_LIB=$L timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --steps 60 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=j['ms']
print('$L'.split('/')[-1].ljust(24), round(j['value']), ' '.join(f'{a}={b:.3f}' for a,b in k.items()))"
  done
done
