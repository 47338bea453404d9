#!/usr/bin/env python3
"""Does the gradient stage's bandwidth depend on WHERE its four output arrays lie relative to each other?  (bench.py's gradient leg came
out at 3.59 or at 4.2-4.4 TB/s from run to run with the same library.)  One big allocation, the arrays carved out of it with a variable
extra offset between them; ms per 1024 frames for each offset."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
import torch, vislam
B, W, H = 1024, 752, 480
p = vislam.default_params(); ctx = vislam.Context(0, p)
d = torch.randint(0, 255, (B, H, W), dtype=torch.uint8, device="cuda")
fe = vislam.gradient_frame_elems(W, H)
big = torch.empty(B * fe * 6 + (64 << 20), dtype=torch.uint8, device="cuda")
base = (big.data_ptr() + (2 << 20) - 1) & ~((2 << 20) - 1)             # 2 MiB aligned
for extra in (0, 256, 4096, 4096 + 256, 65536, 65536 + 4096, 1 << 20, (1 << 20) + 4096 + 256, 3 << 20):
    gray = base; gx = gray + B * fe + extra; gy = gx + 2 * B * fe + extra; g = gy + 2 * B * fe + extra
    gx = (gx + 15) & ~15; gy = (gy + 15) & ~15; g = (g + 15) & ~15
    def run(): ctx.gradient_batch(d.data_ptr(), W, H, W, B, gray, gx, gy, g)
    for _ in range(3): run()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize(); t = (time.perf_counter() - t) / 20
    print("extra offset %8d B between the arrays: %.3f ms per 1024 frames (%.2f TB/s)" % (extra, t * 1e3, 3484110 * B / t / 1e12))
