import sys, os
ROOT="/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT",".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"vi-slam_amd"))
import torch, vislam, bench
dev=torch.device("cuda",0)
q5 = vislam.default_params()
q5.nfeatures, q5.nlevels, q5.w_size, q5.h_size = 8000, 8, 3840, 2160
q5.fy = q5.fx
r=bench.run_leg(dev, 3840, 2160, int(os.environ.get("C5_BATCH", "32")), 2, q5, 0xE0C00005, 8192, int(os.environ.get("C5_STEPS", "5")), 2)
print("config5", round(r["frames_per_s"]), r["kernels_ms_per_step"])
