import sys, os
ROOT="/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT",".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"vi-slam_amd")); sys.path.insert(0, os.path.join(ROOT,"tests"))
import torch, vislam, bench
dev=torch.device("cuda",0)
q3 = vislam.default_params()
q3.nfeatures, q3.nlevels, q3.w_size, q3.h_size = 4000, 4, 1920, 1080
q3.fy = q3.fx
q3.ransac_adaptive, q3.ransac_max_iters, q3.pose_input = 0, 2000, 1
r=bench.run_leg(dev, 1920, 1080, int(os.environ.get("C3_BATCH", "128")), 2, q3, 0xE0C00003, 8192, int(os.environ.get("C3_STEPS", "5")), 2)
print("config3", round(r["frames_per_s"]), r["kernels_ms_per_step"], r["ransac"]["ms_pose_per_step"])
