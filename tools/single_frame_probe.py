#!/usr/bin/env python3
"""Latency of the frame-at-a-time C-ABI sequence (the path main_vi_slamGPU.cpp drives) for the library VISLAM_HIP_LIB selects:
camera_update + orb_detect_compute + good_matches + essential_ransac + recover_pose per frame from pageable host memory.
tools/single_frame_probe.py [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
import numpy as np
import torch  # noqa: F401  (the process-level HIP runtime)
import vislam
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
W, H = 752, 480
p = vislam.default_params(); p.fy = p.fx
ctx = vislam.Context(0, p)
cv = vislam.synth_canvas(2048, 0xE0C00001)
fr = [vislam.synth_frame(cv, t, W, H) for t in range(n + 7)]
names = ("update", "detect", "match", "ransac", "recover")
per = {k: [] for k in names}; tot = []
kp, _ = ctx.orb_detect_compute(fr[0], slot=0)
has_cnt = hasattr(vislam.lib, "vis_debug_counters")
c0 = None
for t in range(1, n + 7):
    if t == 7:
        per = {k: [] for k in names}; tot = []
        c0 = ctx.debug_counters() if has_cnt else None
    a = time.perf_counter(); ctx.camera_update(fr[t])
    b = time.perf_counter(); kc, _d = ctx.orb_detect_compute(fr[t], slot=t & 1)
    c = time.perf_counter(); g, _s = ctx.good_matches((t - 1) & 1, t & 1)
    d = time.perf_counter()
    p1 = np.stack([kp["x"][g["queryIdx"]], kp["y"][g["queryIdx"]]], 1); p2 = np.stack([kc["x"][g["trainIdx"]], kc["y"][g["trainIdx"]]], 1)
    e = time.perf_counter(); E, _m, _ni, _it = ctx.essential_ransac(p1, p2)
    f = time.perf_counter(); ctx.recover_pose(E, p1, p2)
    h = time.perf_counter()
    for k, v in zip(names, (b - a, c - b, d - c, f - e, h - f)): per[k].append(v * 1e3)
    tot.append((d - a + h - e) * 1e3)
    kp = kc
pc = lambda v, q: float(np.percentile(np.array(v), q))
line = f"{os.path.basename(vislam.LIB_PATH):24s} frame p50 {pc(tot, 50):.3f} p95 {pc(tot, 95):.3f} ms | " + " ".join(f"{k} {pc(v, 50):.3f}" for k, v in per.items())
if has_cnt:
    c1 = ctx.debug_counters()
    line += f" | launches/frame {(c1[0] - c0[0]) / n:.1f} waits/frame {(c1[1] - c0[1]) / n:.2f} copies/frame {(c1[2] - c0[2]) / n:.1f}"
print(line)
ctx.close()
