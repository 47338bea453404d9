#!/usr/bin/env python3
"""Workload for profiling the RANSAC-loaded leg alone: S-752, 1024 frames per step, adaptive stop off, 1000 five-point
hypotheses per pair.  rocprofv3 --kernel-trace --stats -- python3 tools/profile_ransac.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
import torch, vislam, bench
B = int(os.environ.get("VIS_PROFILE_BATCH", "1024"))
p = vislam.default_params(); p.nfeatures, p.nlevels, p.w_size, p.h_size = bench.NFEAT, bench.LEVELS, bench.W, bench.H
p.fy = p.fx
p.ransac_adaptive = 0; p.ransac_max_iters = 1000
ctx = vislam.Context(0, p)
stream = bench.Stream(ctx, "cuda:0", bench.W, bench.H, B, 0xE0C00001)
ctx.batch_plan(bench.W, bench.H, bench.W, B)
for i in range(4):
    ctx.batch_run(stream.ptr(0), B, vislam.STAGE_ALL); ctx.batch_sync()
t = ctx.timings()
print("status", ctx.batch_status(), "ms_pose", round(t.ms_pose, 3))
