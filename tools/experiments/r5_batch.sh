#!/bin/bash
# headline against the frames-per-launch choice and the stage mask (same box): tools/experiments/r5_batch.sh
cd $GRAFT_REPO_ROOT
for S in 1 15; do
for B in 512 640 1024 1280 2048 4096; do
  timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --steps 40 --stages $S --batch $B --launches-per-step $((20480 / B)) 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=j['kernels_ms_per_launch']; print('stages $S batch $B', round(j['value']), ' '.join(f'{a[3:]}={b*1024/$B:.3f}' for a,b in k.items()))"
done
done
