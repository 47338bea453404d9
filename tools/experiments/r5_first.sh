#!/bin/bash
# first RANSAC chunk (4 / 8 / 16 hypotheses) on the headline AND on S-752P (whose pairs need 8.7 hypotheses on average); diagnostic build
# (make TAG=_knobs EXTRA=-DVIS_AB_KNOBS lib), same box, alternating
cd $GRAFT_REPO_ROOT
export VISLAM_HIP_LIB=$GRAFT_REPO_ROOT/vi-slam_amd/lib/libvislam_hip_knobs.so
for r in 1 2 3; do for F in 16 8 4; do
  VIS_RANSAC_FIRST=$F timeout -k 10 300 python - <<PY 2>/dev/null
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "vi-slam_amd"))
import torch, bench, vislam
from vislam import dist as vdist
dev = torch.device("cuda:0")
p = vislam.default_params(); p.fy = p.fx
out = []
for par in (False, True):
    r = bench.run_leg(dev, 752, 480, 1024, 2, p, vdist.SINGLE_SEED, 4096, 40, 3, parallax=par)
    out.append("%s %.0f (pose %.3f)" % ("S-752P" if par else "S-752", r["frames_per_s"], r["kernels_ms_per_step"]["ms_pose"]))
print("first$F", " | ".join(out))
PY
done; done
