#!/bin/bash
# k_knn_mfma<1> (100 registers: fits beside four k_fast waves per SIMD) against <2> (172) inside the pipeline; diagnostic build (make TAG=_knobs EXTRA=-DVIS_AB_KNOBS lib)
cd $GRAFT_REPO_ROOT
export VISLAM_HIP_LIB=$GRAFT_REPO_ROOT/vi-slam_amd/lib/libvislam_hip_knobs.so
for r in 1 2 3; do for NC in 2 1; do
  VIS_KNN_NC=$NC timeout -k 10 300 python bench.py --no-legs --no-cpu-baseline --steps 60 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=j['kernels_ms_per_step']
print('nc$NC'.ljust(12), round(j['value']), ' '.join(f'{a[3:]}={b:.3f}' for a,b in k.items()))"
done; done
