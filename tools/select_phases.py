#!/usr/bin/env python3
"""Per-(level, phase) cycles of k_select, thread 0 of every workgroup (diagnostic build: make EXTRA=-DVIS_FAST_PROFILE)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
import torch, vislam, bench
B = 512
p = vislam.default_params(); p.nfeatures, p.nlevels, p.w_size, p.h_size = bench.NFEAT, bench.LEVELS, bench.W, bench.H
ctx = vislam.Context(0, p)
stream = bench.Stream(ctx, "cuda:0", bench.W, bench.H, B, 0xE0C00001)
ctx.batch_plan(bench.W, bench.H, bench.W, B)
out = (C.c_ulonglong * 128)()
f = vislam.lib.vis_debug_select_stamps
for i in range(3):
    ctx.batch_run(stream.ptr(0), B, vislam.STAGE_DETECT); ctx.batch_sync()
    if i == 1: f(out)
assert f(out) == 0
names = ["counts+prefix", "histogram", "cut scan", "gather", "harris", "sort", "keep+write"]
print("cycles per workgroup (100 MHz s_memtime ticks x 24 = core cycles approx.)")
for l in range(8):
    print("level", l, {n: round(out[l * 8 + i] / B) for i, n in enumerate(names)}, "total", round(sum(out[l * 8:l * 8 + 7]) / B))
