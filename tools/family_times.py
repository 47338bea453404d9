#!/usr/bin/env python3
"""Per-family kernel times (HIP events, one batch in flight) of the detect stage only, capacity flags ignored: for timing
experiments with deliberately broken kernels.  VISLAM_HIP_LIB selects the build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
import torch, vislam, bench
B = 1024
p = vislam.default_params(); p.nfeatures, p.nlevels, p.w_size, p.h_size = bench.NFEAT, bench.LEVELS, bench.W, bench.H
p.fy = p.fx
ctx = vislam.Context(0, p)
stream = bench.Stream(ctx, "cuda:0", bench.W, bench.H, B, 0xE0C00001)
ctx.batch_plan(bench.W, bench.H, bench.W, B)
acc = {}
for i in range(8):
    ctx.batch_run(stream.ptr(0), B, vislam.STAGE_DETECT)
    ctx.batch_sync()
    if i >= 2:
        t = ctx.timings()
        for k in ("ms_pyramid", "ms_fast", "ms_select", "ms_describe"):
            acc[k] = acc.get(k, 0.0) + getattr(t, k) / 6
print(os.path.basename(os.environ.get("VISLAM_HIP_LIB", "libvislam_hip.so")), {k: round(v, 3) for k, v in acc.items()})
