#!/usr/bin/env python3
"""Workload for rocprofv3: the gradient stage (Camera::Update half pyramid + computeGradient) over B resident frames."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
import numpy as np, torch, vislam
B = int(os.environ.get("VIS_PROFILE_BATCH", "1024")); W, H = 752, 480
p = vislam.default_params(); ctx = vislam.Context(0, p)
cv = vislam.synth_canvas(4096, 0xE0C00001)
fr = np.empty((B, H, W), np.uint8)
for t in range(B): vislam.synth_frame(cv, t, W, H, 0xE0C00001, out=fr[t])
d = torch.from_numpy(fr).cuda()
fe = vislam.gradient_frame_elems(W, H)
gray = torch.empty(B * fe, dtype=torch.uint8, device="cuda")
gx = torch.empty(B * fe, dtype=torch.int16, device="cuda"); gy = torch.empty_like(gx)
g = torch.empty(B * fe, dtype=torch.uint8, device="cuda")
for _ in range(int(os.environ.get("VIS_PROFILE_STEPS", "4"))):
    ctx.gradient_batch(d.data_ptr(), W, H, W, B, gray.data_ptr(), gx.data_ptr(), gy.data_ptr(), g.data_ptr())
    torch.cuda.synchronize()
# calibration copy of known size (see tools/profile_workload.py)
cal_src = torch.empty(256 << 20, dtype=torch.uint8, device="cuda"); cal_dst = torch.empty_like(cal_src)
torch.cuda.synchronize(); cal_dst.copy_(cal_src); torch.cuda.synchronize()
print("done", B)
ctx.close()
