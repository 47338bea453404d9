"""tools/stage_sets_probe.py -- headline workload (S-752, 1024 frames per launch) with growing stage sets: what the detect chain
runs at alone, and what the matcher / pose / Camera::Update streams cost beside it (frames/s, ms per launch)"""
import sys, os
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, vislam, bench
from vislam import dist as vdist
dev = torch.device("cuda", 0)
p = vislam.default_params()
S = vislam
for name, st in (("detect", S.STAGE_DETECT), ("detect+match", S.STAGE_DETECT | S.STAGE_MATCH), ("detect+match+pose", S.STAGE_ALL),
                 ("update+detect", S.STAGE_DETECT | S.STAGE_UPDATE), ("frame (all)", S.STAGE_FRAME)):
    r = bench.run_leg(dev, 752, 480, 1024, 4, p, vdist.SINGLE_SEED, 4096, 40, 4, stages=st, want_pose=False)
    print(f"{name:20s} {r['frames_per_s']:10.0f} frames/s  {r['ms_per_step']:.3f} ms per launch   kernels {r['kernels_ms_per_step']}")
