#!/usr/bin/env python3
"""Probe: do two independent pipelines on one GPU (separate contexts/streams) deliver more aggregate throughput
than one?  Upper bound for what deeper batch overlap inside one context could gain."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
import numpy as np, torch, vislam
W, H, B = 752, 480, int(os.environ.get("B", "512"))
STAGES = int(os.environ.get("STAGES", str(vislam.STAGE_FRAME)))       # 15 = the headline's stages (Camera::Update included)
p = vislam.default_params(); p.fy = p.fx
cv = vislam.synth_canvas(4096, 0xE0C00001)
fr = np.empty((2 * B, H, W), np.uint8)
for t in range(2 * B):
    vislam.synth_frame(cv, t, W, H, 0xE0C00001, out=fr[t])
d = torch.from_numpy(fr).cuda()
for nctx in (1, 2, 1, 2):
    ctxs = [vislam.Context(0, p) for _ in range(nctx)]
    for c in ctxs: c.batch_plan(W, H, W, B)
    def run(steps):
        for i in range(steps):
            for c in ctxs: c.batch_run(d.data_ptr() + (i & 1) * B * W * H, B, STAGES)
        for c in ctxs: c.batch_sync()
    run(4); torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 30; run(K); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"contexts={nctx}: {K * nctx * B / dt:.0f} frames/s", flush=True)
    for c in ctxs: c.close()
