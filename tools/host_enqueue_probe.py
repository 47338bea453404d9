#!/usr/bin/env python3
"""Host-side duration of every vis_batch_run call of a pipelined run (no per-step sync): does the enqueue run ahead of the GPU, or does
a call block?  tools/host_enqueue_probe.py [stages] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
import torch, vislam, bench
stages = int(sys.argv[1]) if len(sys.argv) > 1 else vislam.STAGE_FRAME
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
B = 1024
p = vislam.default_params(); p.nfeatures, p.nlevels, p.w_size, p.h_size = bench.NFEAT, bench.LEVELS, bench.W, bench.H
p.fy = p.fx
ctx = vislam.Context(0, p)
stream = bench.Stream(ctx, "cuda:0", bench.W, bench.H, 2 * B, 0xE0C00001)
ctx.batch_plan(bench.W, bench.H, bench.W, B)
for i in range(5):
    ctx.batch_run(stream.ptr((i % 2) * B), B, stages)
ctx.batch_sync(); torch.cuda.synchronize()
t0 = time.perf_counter(); ts = []
for i in range(steps):
    a = time.perf_counter()
    ctx.batch_run(stream.ptr((i % 2) * B), B, stages)
    ts.append((a - t0, time.perf_counter() - a))
ctx.batch_sync(); torch.cuda.synchronize()
t = (time.perf_counter() - t0) / steps
print(f"stages {stages}: {t * 1e3:.3f} ms per step")
print("call start (ms) / host duration (ms):", " ".join(f"{a * 1e3:.2f}/{d * 1e3:.2f}" for a, d in ts))
