"""tools/fixed1000_probe.py -- the fixed-1000 leg of bench.py alone (S-752, ransac_adaptive = 0: 1000 five-point hypotheses per pair),
for rocprofv3 --kernel-trace --stats / --pmc runs of the pose kernels under load"""
import sys, os
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, vislam, bench
from vislam import dist as vdist
dev = torch.device("cuda", 0)
q1 = vislam.default_params()
q1.ransac_adaptive = 0
r = bench.run_leg(dev, 752, 480, int(os.environ.get("F1000_BATCH", "1024")), 2, q1, vdist.SINGLE_SEED, 4096, 6, 2)
print("fixed1000", round(r["frames_per_s"]), r["kernels_ms_per_step"], r["ransac"]["ms_pose_per_step"])
