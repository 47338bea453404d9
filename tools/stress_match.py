#!/usr/bin/env python3
"""tools/stress_match.py [SECONDS] [SEED] -- randomised parity of the host-pointer matcher entry points and VISystem::F2FRansac on the
GPU box: vis_bf_knn2_hamming_host on random descriptor sets (1 ... 9000 rows each side, with duplicated rows, all-zero / all-one rows
and near-duplicates so that ties and the ends of the distance range occur), vis_good_matches_host on random keypoints with those
tables, vis_f2f_ransac on random two-view problems -- 2-NN tables, symmetric / good matches and the winning count bit for bit, the
translation within 1e-6.  Exit code 1 on any failure.  Not part of the test suite (unbounded run time)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import vislam  # noqa: E402
import oracle_bind as orc  # noqa: E402
from test_pose_gpu import _f2f_inputs  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 31
max_cases = int(sys.argv[3]) if len(sys.argv) > 3 else None     # a CASE COUNT bounds the run (the same cases on every box); the seconds are then only a guard
rng = np.random.default_rng(seed0)
ctx = vislam.Context(0)
KP = vislam.KEYPOINT_DTYPE


def descs(n):
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    if n > 4 and rng.random() < 0.5:                              # duplicates, extremes, near-duplicates (Hamming 1 ... 3)
        k = int(rng.integers(1, max(2, n // 3)))
        src = rng.integers(0, n, k); dst = rng.integers(0, n, k)
        d[dst] = d[src]
        flip = rng.integers(0, n, k // 2 + 1)
        d[flip, rng.integers(0, 32, len(flip))] ^= (1 << rng.integers(0, 8, len(flip))).astype(np.uint8)
        d[rng.integers(0, n, 2)] = 0
        d[rng.integers(0, n, 2)] = 255
    return d


t_end = time.time() + budget
runs = fails = 0
per_kind = [0, 0, 0]
while time.time() < t_end and (max_cases is None or runs < max_cases):
    kind = int(rng.integers(0, 3))
    per_kind[kind] += 1
    what = "?"
    try:
        if kind == 0:
            n1 = int(rng.choice([1, 2, 3, 31, 32, 33, 63, 64, 65, 255, 256, 257, 1000, 1000, 1023, 1025, 4000, 8000, 9000]))
            n2 = int(rng.choice([1, 2, 7, 31, 33, 64, 65, 129, 500, 1000, 1000, 1024, 4000, 8000]))
            if n1 * n2 > 40_000_000:
                n2 = 1000
            what = f"knn {n1} x {n2}"
            d1, d2 = descs(n1), descs(n2)
            g12, g21 = ctx.bf_knn2_hamming_host(d1, d2)
            o12, o21 = orc.knn2_hamming(d1, d2)
            ok = g12.tobytes() == o12.tobytes() and g21.tobytes() == o21.tobytes()
        elif kind == 1:
            n1, n2 = int(rng.integers(1, 1500)), int(rng.integers(1, 1500))
            what = f"filters {n1} x {n2}"
            p = vislam.default_params()
            p.w_size, p.h_size = int(rng.choice([752, 640, 320])), int(rng.choice([480, 480, 240]))
            p.n_cells = int(rng.choice([49, 49, 16, 100, 50]))
            p.sym_mode = int(rng.integers(0, 2))
            p.ratio = float(rng.choice([0.8, 0.8, 0.6, 0.95]))
            ctx.set_params(p)
            k1, k2 = np.zeros(n1, KP), np.zeros(n2, KP)
            for k, n in ((k1, n1), (k2, n2)):
                k["x"] = rng.uniform(0, p.w_size - 0.01, n).astype(np.float32); k["y"] = rng.uniform(0, p.h_size - 0.01, n).astype(np.float32)
                if rng.random() < 0.3:
                    k["y"] = np.round(k["y"])                       # equal y: the sort's tie rule
            d1, d2 = descs(n1), descs(n2)
            if rng.random() < 0.5 and n1 > 10 and n2 > 10:        # correlated sets: most rows have a close partner, so the filters keep something
                m = min(n1, n2)
                d2[:m] = d1[:m]
                d2[:m, 0] ^= rng.integers(0, 4, m).astype(np.uint8)
            o12, o21 = orc.knn2_hamming(d1, d2)
            good, sym = ctx.good_matches_host(k1, k2, o12, o21)
            og, osym = orc.good_matches(p, k1, k2, o12, o21)
            ok = good.tobytes() == og.tobytes() and sym.tobytes() == osym.tobytes()
        else:
            m = int(rng.choice([2, 3, 15, 40, 40, 120, 400]))
            what = f"f2f {m}"
            p = vislam.default_params()
            p.f2f_threshold = float(rng.choice([250.0, 370.0, 370.0, 600.0]))
            ctx.set_params(p)
            sd = int(rng.integers(1, 1 << 30))
            a, b, rot = _f2f_inputs(vislam, m, sd, float(rng.choice([0.0, 0.2, 0.5])), float(rng.choice([0.0, 0.3, 1.0])))
            idx = rng.integers(0, max(m - 1, 1), (int(rng.choice([1, 100, 1000])), 2)).astype(np.int32)
            got, cg = ctx.f2f_ransac(a, b, rot, idx, 0.37)
            ref, co = orc.f2f_ransac(p, a, b, rot, idx, 0.37)
            ok = cg == co and np.abs(got - ref).max() <= 1e-6
            what += f" seed {sd} counts {(cg, co)}"
    except Exception as e:
        ok = False
        what += " raised " + repr(e)[:200]
    runs += 1
    if not ok:
        fails += 1
        print("FAIL", what, flush=True)
print(f"stress_match: {runs} cases (knn {per_kind[0]}, filters {per_kind[1]}, f2f {per_kind[2]}), {fails} failures, seed {seed0}")
ctx.close()
sys.exit(1 if fails else 0)
