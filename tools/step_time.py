#!/usr/bin/env python3
"""Wall-clock time per pipelined step (no per-step sync), for scheduling experiments: tools/step_time.py [stages] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
import torch, vislam, bench
stages = int(sys.argv[1]) if len(sys.argv) > 1 else vislam.STAGE_FRAME
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
B = 1024
p = vislam.default_params(); p.nfeatures, p.nlevels, p.w_size, p.h_size = bench.NFEAT, bench.LEVELS, bench.W, bench.H
p.fy = p.fx
ctx = vislam.Context(0, p)
stream = bench.Stream(ctx, "cuda:0", bench.W, bench.H, 2 * B, 0xE0C00001)
ctx.batch_plan(bench.W, bench.H, bench.W, B)
for i in range(5):
    ctx.batch_run(stream.ptr((i % 2) * B), B, stages)
ctx.batch_sync(); torch.cuda.synchronize()
t = time.perf_counter()
for i in range(steps):
    ctx.batch_run(stream.ptr((i % 2) * B), B, stages)
ctx.batch_sync(); torch.cuda.synchronize()
t = (time.perf_counter() - t) / steps
print(f"stages {stages}: {t * 1e3:.3f} ms per step, {B / t:.0f} frames/s  (status {ctx.batch_status()})")
