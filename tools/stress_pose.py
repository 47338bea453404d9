#!/usr/bin/env python3
"""tools/stress_pose.py [SECONDS] [SEED] -- randomised parity of the pose stage on the GPU box: random two-view problems (5 ... 6000
correspondences, outlier rates, noise, thresholds, focal lengths, adaptive and fixed iteration counts up to 2000) through
vis_find_essential / vis_recover_pose against the CPU oracle: inlier mask, inlier count and iterations identical, E / R / t within 1e-9
(the tolerance of tests/test_pose_gpu.py).  Prints one line per failure and a summary; exit code 1 on any failure.  Not part of the
test suite (unbounded run time); tests/test_pose_gpu.py holds the fixed cases."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import vislam  # noqa: E402
import oracle_bind as orc  # noqa: E402
from test_pose_gpu import two_view, _cmpE  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
max_cases = int(sys.argv[3]) if len(sys.argv) > 3 else None     # a CASE COUNT bounds the run (the same cases on every box); the seconds are then only a guard
rng = np.random.default_rng(seed0)
ctx = vislam.Context(0)
t_end = time.time() + budget
runs = fails = 0
big = 0
while time.time() < t_end and (max_cases is None or runs < max_cases):
    p = vislam.default_params()
    p.fx = float(rng.choice([150.0, 458.654, 458.654, 900.0])); p.fy = p.fx
    p.ransac_threshold = float(rng.choice([0.25, 1.0, 1.0, 3.0]))
    p.ransac_prob = float(rng.choice([0.9, 0.999, 0.999]))
    adaptive = int(rng.random() < 0.5)
    p.ransac_adaptive = adaptive
    p.ransac_max_iters = int(rng.choice([16, 17, 64, 100, 300, 1000, 2000]))
    n = int(rng.choice([5, 6, 7, 20, 49, 49, 120, 255, 256, 257, 300, 700, 1025, 3100, 6000]))
    if not adaptive and n > 1000 and p.ransac_max_iters > 300:
        big += 1
        if big % 4:                                             # the oracle needs seconds for these: one in four
            p.ransac_max_iters = 300
    outl, noise = float(rng.choice([0.0, 0.1, 0.3, 0.6])), float(rng.choice([0.0, 0.2, 0.5, 1.5]))
    sd = int(rng.integers(1, 1 << 30))
    x1, x2, R, t = two_view(n, sd, outl, noise)
    ctx.set_params(p)
    E, mask, ninl, iters = ctx.essential_ransac(x1, x2)
    oE, omask, oninl, oiters = orc.essential_ransac(p, x1, x2)
    ok = (ninl, iters) == (oninl, oiters) and bool((mask == omask).all())
    if ok and oninl > 0:
        ok = _cmpE(E, oE) <= 1e-9
        if ok:
            Rg, tg, ng = ctx.recover_pose(oE, x1, x2)
            Ro, to, no = orc.recover_pose(p, oE, x1, x2)
            ok = ng == no and np.abs(Rg - Ro).max() <= 1e-9 and np.abs(tg - to).max() <= 1e-9
    runs += 1
    if not ok:
        fails += 1
        print("FAIL", dict(n=n, seed=sd, outl=outl, noise=noise, fx=p.fx, thr=p.ransac_threshold, prob=p.ransac_prob, adaptive=adaptive,
                           iters=p.ransac_max_iters), "gpu", (ninl, iters), "oracle", (oninl, oiters), flush=True)
print(f"stress_pose: {runs} problems, {fails} failures, seed {seed0}")
ctx.close()
sys.exit(1 if fails else 0)
