#!/usr/bin/env python3
"""Soak run of the batched pipeline: the same ring of resident frames is pushed through vis_batch_run launch after launch for SECONDS
seconds, without a host sync between the launches of a pass (the results of every launch are queued into pinned memory behind it, as
bench.py's D2H leg does); every pass must reproduce pass 0's pose records, good-match counts and good matches byte for byte.
A missing or misplaced event between the four streams (detect chain running ahead into a record set the matcher still reads, results
copied before the pose stage wrote them, ...) shows up as a pass that differs -- the timing-dependent failures a short test cannot provoke.

    python tools/soak_pipeline.py SECONDS [BATCH] [LAUNCHES_PER_PASS] [parallax | main]

"main": the GPU main's sequence instead (detect + matcher + Scharr gradients on the side stream + vis_batch_align on the pose stream, the
plan's two gradient sets used in turn); the alignment records of every launch are hashed.
"""
import ctypes as C
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vislam  # noqa: E402

W, H = 752, 480
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 12
parallax = len(sys.argv) > 4 and sys.argv[4] == "parallax"
main_seq = len(sys.argv) > 4 and sys.argv[4] == "main"
max_passes = int(sys.argv[5]) if len(sys.argv) > 5 else None     # a PASS COUNT bounds the run; the seconds are then only a guard

p = vislam.default_params(); p.fy = p.fx
ctx = vislam.Context(0, p)
dev = torch.device("cuda:0")
seed = 0xE0C00001
canvas = vislam.synth_canvas(4096, seed)
d_canvas = torch.from_numpy(canvas).to(dev)
frames = torch.empty((B * Q, H, W), dtype=torch.uint8, device=dev)
for t0 in range(0, B * Q, 256):
    n = min(256, B * Q - t0)
    ctx.synth_frames_device(d_canvas.data_ptr(), 4096, seed, t0, n, W, H, W, frames.data_ptr() + t0 * W * H, parallax)
torch.cuda.synchronize()
ctx.batch_plan(W, H, W, B)
root2 = int(np.floor(np.sqrt(p.n_cells))) ** 2
hp = [torch.zeros(B * C.sizeof(vislam.PoseResult), dtype=torch.uint8).pin_memory() for _ in range(Q)]
hg = [torch.zeros(B * root2 * 16, dtype=torch.uint8).pin_memory() for _ in range(Q)]
hn = [torch.zeros(B, dtype=torch.int32).pin_memory() for _ in range(Q)]
asz = C.sizeof(vislam.AlignResult)
aout = [torch.zeros(B * asz, dtype=torch.uint8, device=dev) for _ in range(Q)] if main_seq else None
apar = vislam.default_align_params()


def one_pass():
    ctx.batch_reset()                                  # a pass is a fresh stream: frame 0 has no predecessor
    if main_seq:
        for q in range(Q):
            d = frames.data_ptr() + q * B * W * H
            ctx.batch_run(d, B, vislam.STAGE_DETECT | vislam.STAGE_MATCH | vislam.STAGE_GRADIENT)
            ctx.batch_align(apar, d, B, 0, 0, 0, 0, aout[q].data_ptr())
        ctx.batch_sync()
        if ctx.batch_status() != 0:
            raise RuntimeError("device capacity flag set")
        h = hashlib.sha256()
        for q in range(Q):
            h.update(aout[q].cpu().numpy().tobytes())
        return h.hexdigest()
    for q in range(Q):
        ctx.batch_run(frames.data_ptr() + q * B * W * H, B)
        ctx.batch_results_async(B, hp[q].data_ptr(), hg[q].data_ptr(), hn[q].data_ptr())
    ctx.batch_sync()
    if ctx.batch_status() != 0:
        raise RuntimeError("device capacity flag set")
    h = hashlib.sha256()
    for q in range(Q):
        ng = hn[q].numpy()
        h.update(hp[q].numpy().tobytes()); h.update(ng.tobytes())
        good = hg[q].numpy().reshape(B, root2, 16)
        for t in range(B):                              # rows are dense up to the count; the rest of a row is not written
            h.update(good[t, :ng[t]].tobytes())
    return h.hexdigest()


ref = one_pass()
t_end = time.time() + seconds
passes, bad, t_print = 1, 0, time.time()
while time.time() < t_end and (max_passes is None or passes < max_passes):
    d = one_pass()
    passes += 1
    if d != ref:
        bad += 1
        print(f"pass {passes}: DIFFERENT {d[:16]} != {ref[:16]}", flush=True)
    if time.time() - t_print > 30:
        print(f"... {passes} passes, {bad} different", flush=True); t_print = time.time()
ctx.close()
print(f"soak_pipeline: {passes} passes of {Q} launches x {B} frames ({'S-752P' if parallax else ('S-752, main sequence' if main_seq else 'S-752')}), {passes * Q * B} frames, {bad} passes differ, reference {ref[:16]}")
sys.exit(1 if bad else 0)
