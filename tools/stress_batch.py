#!/usr/bin/env python3
"""tools/stress_batch.py [SECONDS] [SEED] -- randomised parity of the BATCH pipeline (vis_batch_run: detect | match | pose over streams,
carried last frame, speculative FAST thresholds, work lists) against the oracle's per-frame pipeline: random stream contents (crops
with and without parallax, contrast changes mid-stream, noise frames in between), random image sizes, ORB / match / RANSAC
parameters (adaptive and fixed iterations), random cuts of the stream into batches on ONE context.
Per frame: keypoint count, symmetric / good match counts, inliers, iterations identical; E / R / t within the tolerances of
tests/test_pose_gpu.py.  Exit code 1 on any failure.  Not part of the test suite (unbounded run time)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vislam  # noqa: E402
import oracle_bind as orc  # noqa: E402
from test_pose_gpu import _cmpE  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 777
max_cases = int(sys.argv[3]) if len(sys.argv) > 3 else None     # a CASE COUNT bounds the run (the same cases on every box); the seconds are then only a guard
rng = np.random.default_rng(seed0)
canvas = vislam.synth_canvas(2048, 0xE0C00001)
t_end = time.time() + budget
runs = fails = frames_total = 0
while time.time() < t_end and (max_cases is None or runs < max_cases):
    w, h = [(752, 480), (752, 480), (320, 240), (641, 479), (500, 375)][int(rng.integers(0, 5))]
    p = vislam.default_params()
    p.w_size, p.h_size = w, h
    p.fy = p.fx
    p.nfeatures = int(rng.choice([200, 500, 1000, 1000]))
    p.nlevels = int(rng.choice([3, 5, 8, 8]))
    p.sym_mode = int(rng.integers(0, 2))
    p.ransac_adaptive = int(rng.random() < 0.6)
    p.ransac_max_iters = int(rng.choice([16, 40, 100, 300])) if not p.ransac_adaptive else int(rng.choice([100, 1000]))
    p.ransac_threshold = float(rng.choice([0.5, 1.0, 1.0, 2.0]))
    # (pose_input stays VIS_POSE_GRID: the oracle's per-frame pipeline models the reference's, which has no other)
    n = int(rng.integers(3, 28))
    if rng.random() < 0.05:                                      # a long stream with the adaptive stop off: a work list beyond the resident grid of
        w, h = 320, 240                                          # the roots kernel (its items are claimed from a counter), ~10 s of oracle time
        p.w_size, p.h_size, p.nfeatures, p.nlevels = w, h, 200, 5
        p.ransac_adaptive, p.ransac_max_iters = 0, 1000
        n = int(rng.integers(150, 200))
    par = bool(rng.integers(0, 2))
    t0 = int(rng.integers(0, 300)); step = int(rng.choice([1, 1, 2, 5]))
    frames = []
    for i in range(n):
        f = vislam.synth_frame(canvas, t0 + step * i, w, h, parallax=par)
        r = rng.random()
        if r < 0.08:
            f = rng.integers(0, 256, (h, w), dtype=np.uint8)      # a noise frame: mispredicted thresholds, hardly any match
        elif r < 0.2:
            f = (f // 4 + 90).astype(np.uint8)                    # quarter contrast
        frames.append(f)
    frames = np.stack(frames)
    stride = (w + 3) & ~3                                         # the batch API wants rows on 4-byte boundaries (641 -> 644, padding = noise)
    padded = rng.integers(0, 256, (n, h, stride), dtype=np.uint8)
    padded[:, :, :w] = frames
    dev = torch.from_numpy(padded).cuda()
    bmax = int(rng.integers(1, n + 1))
    what = "plan"; print("cfg", dict(w=w, h=h, n=n, bmax=bmax, nfeat=p.nfeatures, levels=p.nlevels, adaptive=p.ransac_adaptive, iters=p.ransac_max_iters), flush=True) if os.environ.get("STRESS_VERBOSE") else None
    try:
        c = vislam.Context(0, p)
        c.batch_plan(w, h, stride, bmax)
        got = []
        i = 0
        while i < n:
            nb = int(rng.integers(1, bmax + 1)); nb = min(nb, n - i)
            what = f"batch_run({i}, {nb})"
            c.batch_run(dev.data_ptr() + i * stride * h, nb); c.batch_sync()
            if c.batch_status() != 0:
                raise RuntimeError("capacity flag")
            for t in range(nb):
                g, nsym = c.batch_matches(t)
                pose = c.batch_pose(t)
                got.append((len(c.batch_keypoints(t)[0]), nsym, len(g), dict((k, np.array(pose[k]).copy()) for k in ("n_inliers", "iters_run", "n_pose_good", "E", "R", "t"))))
            i += nb
        c.close()
        what = "compare"
        prev = None
        ok = True
        for t in range(n):
            okp, od, r = orc.pipeline_frame(p, frames[t], prev)
            prev = (okp, od)
            nk, nsym, ng, pose = got[t]
            good = (nk == len(okp) and nsym == r.n_sym and ng == r.n_good and int(pose["n_inliers"]) == r.n_inliers and int(pose["iters_run"]) == r.iters_run)
            if good and r.n_inliers:
                good = (_cmpE(pose["E"], np.array(r.E).reshape(3, 3)) <= 1e-9 and int(pose["n_pose_good"]) == r.n_pose_good
                        and np.abs(pose["R"] - np.array(r.R).reshape(3, 3)).max() <= 1e-7 and np.abs(pose["t"] - np.array(r.t)).max() <= 1e-7)
            if not good:
                ok = False
                what = f"frame {t}: gpu {(nk, nsym, ng, int(pose['n_inliers']), int(pose['iters_run']))} oracle {(len(okp), r.n_sym, r.n_good, r.n_inliers, r.iters_run)}"
                break
    except Exception as e:
        ok = False
        what += " raised " + repr(e)[:200]
    runs += 1; frames_total += n
    if not ok:
        fails += 1
        print("FAIL", what, dict(w=w, h=h, n=n, bmax=bmax, nfeat=p.nfeatures, levels=p.nlevels, sym=p.sym_mode, adaptive=p.ransac_adaptive, iters=p.ransac_max_iters,
                                 thr=p.ransac_threshold, parallax=par, t0=t0, step=step), flush=True)
print(f"stress_batch: {runs} streams ({frames_total} frames), {fails} failures, seed {seed0}")
sys.exit(1 if fails else 0)
