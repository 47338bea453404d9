#!/usr/bin/env python3
"""Wall-clock time per pipelined step (no per-step sync), for scheduling experiments: tools/step_time.py [stages] [steps] [d2h]
("d2h": the results of every step are queued into pinned host memory behind it, as bench.py's legs.s752_results_d2h does)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
import torch, vislam, bench
stages = int(sys.argv[1]) if len(sys.argv) > 1 else vislam.STAGE_FRAME
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
B = 1024
d2h = len(sys.argv) > 3 and sys.argv[3] == "d2h"
p = vislam.default_params(); p.nfeatures, p.nlevels, p.w_size, p.h_size = bench.NFEAT, bench.LEVELS, bench.W, bench.H
p.fy = p.fx
ctx = vislam.Context(0, p)
stream = bench.Stream(ctx, "cuda:0", bench.W, bench.H, 2 * B, 0xE0C00001)
ctx.batch_plan(bench.W, bench.H, bench.W, B)
if d2h:
    import ctypes as C
    hp = torch.zeros(B * C.sizeof(vislam.PoseResult), dtype=torch.uint8).pin_memory(); hg = torch.zeros(B * 49 * 16, dtype=torch.uint8).pin_memory(); hn = torch.zeros(B, dtype=torch.int32).pin_memory()
def run(i):
    ctx.batch_run(stream.ptr((i % 2) * B), B, stages)
    if d2h:
        if os.environ.get("D2H_N") == "1":
            ctx.batch_results_async(B, hp.data_ptr(), None, None)
        else:
            ctx.batch_results_async(B, hp.data_ptr(), hg.data_ptr(), hn.data_ptr())
for i in range(5):
    run(i)
ctx.batch_sync(); torch.cuda.synchronize()
t = time.perf_counter()
for i in range(steps):
    run(i)
ctx.batch_sync(); torch.cuda.synchronize()
t = (time.perf_counter() - t) / steps
print(f"stages {stages}: {t * 1e3:.3f} ms per step, {B / t:.0f} frames/s  (status {ctx.batch_status()})")
