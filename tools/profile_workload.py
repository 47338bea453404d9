#!/usr/bin/env python3
"""Workload for rocprofv3 PMC passes: a few batched steps of the bench configuration (B frames of S-752)
plus one device-to-device copy of known size that calibrates FETCH_SIZE/WRITE_SIZE on this box
(MI355X_MICROARCH.md 'HBM': gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x; other access
widths must be calibrated on a known byte count).  Run as:
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 tools/profile_workload.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vislam  # noqa: E402

B = int(os.environ.get("VIS_PROFILE_BATCH", "256"))
STEPS = int(os.environ.get("VIS_PROFILE_STEPS", "3"))
# VIS_PROFILE_CONFIG: headline (BASELINE configs[1]) | c3 (configs[2]: 1920x1080, 4 levels, 4000 kps, 2000 fixed RANSAC iterations on the
# symmetric matches) | c5 (configs[4]: 3840x2160, 8 levels, 8000 kps) -- the same parameters bench.py's legs run (every
# configuration is quoted with counters of ITS OWN workload)
CFG = os.environ.get("VIS_PROFILE_CONFIG", "headline")
p = vislam.default_params()
W, H, DIM, SEED = 752, 480, 4096, 0xE0C00001
if CFG == "c3":
    W, H, DIM, SEED = 1920, 1080, 8192, 0xE0C00003
    p.nfeatures, p.nlevels, p.w_size, p.h_size = 4000, 4, W, H
    p.ransac_adaptive, p.ransac_max_iters, p.pose_input = 0, 2000, 1
elif CFG == "c5":
    W, H, DIM, SEED = 3840, 2160, 8192, 0xE0C00005
    p.nfeatures, p.nlevels, p.w_size, p.h_size = 8000, 8, W, H
elif CFG != "headline":
    raise SystemExit(f"unknown VIS_PROFILE_CONFIG {CFG}")
p.fy = p.fx
ctx = vislam.Context(0, p)
cv = vislam.synth_canvas(DIM, SEED)
dcv = torch.from_numpy(cv).cuda()
d = torch.empty((B, H, W), dtype=torch.uint8, device="cuda")
GEN = max(1, min(256, (256 * 752 * 480) // (W * H)))
for t0 in range(0, B, GEN):            # device-side generator: byte-identical to the host one (tests/test_synth_gpu.py)
    ctx.synth_frames_device(dcv.data_ptr(), DIM, SEED, t0, min(GEN, B - t0), W, H, W, d.data_ptr() + t0 * W * H)
torch.cuda.synchronize()
ctx.batch_plan(W, H, W, B)
# flush the 256 MiB Infinity Cache between steps with a 1 GiB fill so every step reads its frames from HBM
junk = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
cal_src = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
cal_dst = torch.empty_like(cal_src)
# two untimed warm-up steps: the FAST threshold prediction of the batched path starts with the second batch of a stream
for i in range(2):
    ctx.batch_run(d.data_ptr(), B, vislam.STAGE_FRAME)
    ctx.batch_sync()
for i in range(STEPS):
    junk.fill_(i)
    torch.cuda.synchronize()
    ctx.batch_run(d.data_ptr(), B, vislam.STAGE_FRAME)
    ctx.batch_sync()
junk.fill_(7)
torch.cuda.synchronize()
cal_dst.copy_(cal_src)          # calibration: 256 MiB read + 256 MiB written
torch.cuda.synchronize()
print("status", ctx.batch_status(), "B", B, "config", CFG)
ctx.close()
