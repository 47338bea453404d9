#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do for F in 0 1 2; do
  VIS_UPDATE_FORK=$F timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --steps 60 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=j['kernels_ms_per_step']
print('fork $F', round(j['value']), ' '.join(f'{a[3:]}={b:.3f}' for a,b in k.items()))"
done; done
