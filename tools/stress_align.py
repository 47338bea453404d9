#!/usr/bin/env python3
"""tools/stress_align.py [SECONDS] [SEED] -- randomised parity of Camera::computeGradient and VISystem::EstimatePoseFeatures (rows N2 /
N4 of SURVEY 8(f)) on the GPU box: random image sizes (any size >= 64), shifts, candidate counts, gradient divisors (step sizes),
level ranges, iteration limits, intrinsics and initial poses through vis_compute_gradient / vis_estimate_pose_features against the CPU
oracle -- gradients (int16 x/y, blended u8) and alignment results (iterations, residual counts, errors, the 7 pose floats) bit for
bit.  Exit code 1 on any failure.  Not part of the test suite (unbounded run time); tests/test_align_gpu.py, test_gradient_gpu.py hold
the fixed cases."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import vislam  # noqa: E402
import oracle_bind as orc  # noqa: E402
import align_cases  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 99
max_cases = int(sys.argv[3]) if len(sys.argv) > 3 else None     # a CASE COUNT bounds the run (the same cases on every box); the seconds are then only a guard
rng = np.random.default_rng(seed0)
ctx = vislam.Context(0)
canvas = vislam.synth_canvas(2048, 0xE0C00001)
t_end = time.time() + budget
runs = fails = 0
while time.time() < t_end and (max_cases is None or runs < max_cases):
    w = int(rng.choice([752, 752, 320, 641, 500, 270, 1080, 150])); h = int(rng.choice([480, 480, 240, 479, 375, 150, 540, 110]))
    dx, dy = int(rng.integers(-8, 9)), int(rng.integers(-6, 7))
    n = int(rng.choice([5, 12, 30, 49, 49, 120]))
    div = int(rng.choice([1, 1, 8, 16, 96, 400]))
    what = "case"
    try:
        c = align_cases.case(vislam, orc, canvas, w=w, h=h, dx=dx, dy=dy, n=n, grad_div=div, t=int(rng.integers(0, 200)))
        what = "gradient"
        scale = int(rng.choice([3, 3, 1, 5]))
        gx, gy, g = ctx.compute_gradient(c["f0"], scale)
        ok = True
        lv = orc.half_pyramid(c["f0"])
        for l in range(5):
            a, b, bl = orc.scharr_gradient(lv[l], scale)
            ok = ok and np.array_equal(gx[l], a) and np.array_equal(gy[l], b) and np.array_equal(g[l], bl)
        if ok:
            what = "alignment"
            ap = vislam.default_align_params(); oap = orc.default_align_params()
            first = int(rng.integers(0, 5)); last = int(rng.integers(0, first + 1)); iters = int(rng.choice([1, 3, 10, 10, 25]))
            f = float(rng.choice([458.654, 300.0, 150.0]))
            for q in (ap, oap):
                q.first_level, q.last_level, q.max_iterations = first, last, iters
                q.fx, q.fy, q.cx, q.cy = f, f, w / 2.0, h / 2.0
            init = None
            if rng.random() < 0.4:
                init = orc.se3_exp([float(x) for x in rng.normal(0, 1, 6) * np.array([0.01, 0.01, 0.005, 1e-3, 1e-3, 1e-3])])
            got = ctx.estimate_pose_features(ap, w, h, c["gray1"], c["gray2"], c["gx"], c["gy"], c["cand"], init)
            ref = orc.estimate_pose_features(oap, w, h, c["gray1"], c["gray2"], c["gx"], c["gy"], c["cand"], init)
            ta, tb = align_cases.result_tuple(got), align_cases.result_tuple(ref)
            ok = (ta[0] == tb[0] and ta[1] == tb[1] and ta[3] == tb[3] and np.array_equal(np.array(ta[2], np.float32), np.array(tb[2], np.float32))
                  and np.array_equal(np.array(ta[4], np.float32), np.array(tb[4], np.float32)) and np.array_equal(np.array(ta[5], np.float32), np.array(tb[5], np.float32)))
    except Exception as e:
        ok = False
        what += " raised " + repr(e)[:200]
    runs += 1
    if not ok:
        fails += 1
        print("FAIL at", what, dict(w=w, h=h, dx=dx, dy=dy, n=n, div=div), flush=True)
print(f"stress_align: {runs} cases, {fails} failures, seed {seed0}")
ctx.close()
sys.exit(1 if fails else 0)
