#!/usr/bin/env python3
"""Wall-clock time per pipelined step of the GPU main's per-frame sequence (bench.py legs.gpu_main_sequence: half pyramid + detect +
matcher + gradients + alignment), for timeline / scheduling work: tools/main_sequence_time.py [steps]"""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
import torch, vislam, bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 1024
p = vislam.default_params(); p.nfeatures, p.nlevels, p.w_size, p.h_size = bench.NFEAT, bench.LEVELS, bench.W, bench.H
p.fy = p.fx
ctx = vislam.Context(0, p)
stream = bench.Stream(ctx, "cuda:0", bench.W, bench.H, 2 * B, 0xE0C00001)
ctx.batch_plan(bench.W, bench.H, bench.W, B)
outb = torch.empty(B * C.sizeof(vislam.AlignResult), dtype=torch.uint8, device="cuda:0")
apar = vislam.default_align_params()
def main_step(i):
    d = stream.ptr((i % 2) * B)
    ctx.batch_run(d, B, vislam.STAGE_DETECT | vislam.STAGE_MATCH | vislam.STAGE_GRADIENT)
    ctx.batch_align(apar, d, B, 0, 0, 0, 0, outb.data_ptr())
for i in range(4):
    main_step(i)
ctx.batch_sync(); torch.cuda.synchronize()
t = time.perf_counter()
for i in range(steps):
    main_step(4 + i)
ctx.batch_sync(); torch.cuda.synchronize()
t = (time.perf_counter() - t) / steps
print(f"main sequence: {t * 1e3:.3f} ms per step, {B / t:.0f} frames/s  (status {ctx.batch_status()})")
