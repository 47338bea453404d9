"""CPU: hand-checkable unit tests of the oracle (the reference ships no tests; these follow the plan in
SURVEY.md section 4: FAST on synthetic corners, Hamming of known bit patterns, ratio/symmetry/grid
filter on hand-made DMatch lists, essential matrix from a known R,t)."""
import math

import numpy as np
import pytest


def P(vislam, **kw):
    p = vislam.default_params()
    for k, v in kw.items():
        setattr(p, k, v)
    return p


# ---------------------------------------------------------------- geometry (SURVEY 8(a) a4 numbers)
def test_level_sizes_and_quotas(vislam, orc):
    ws, hs, sc, q = orc.level_geometry(P(vislam), 752, 480)
    assert list(ws) == [752, 627, 522, 435, 363, 302, 252, 210]
    assert list(hs) == [480, 400, 333, 278, 231, 193, 161, 134]
    assert list(q) == [217, 181, 151, 126, 105, 87, 73, 60]
    assert list(orc.level_geometry(P(vislam, nfeatures=200), 752, 480)[3]) == [43, 36, 30, 25, 21, 17, 15, 13]
    assert list(orc.level_geometry(P(vislam, nfeatures=4000, nlevels=4), 1920, 1080)[3]) == [1288, 1073, 894, 745]
    assert list(orc.level_geometry(P(vislam, nfeatures=8000), 3840, 2160)[3]) == [1737, 1448, 1207, 1005, 838, 698, 582, 485]
    assert int(sum(int(a) * int(b) for a, b in zip(ws, hs))) == 1117367      # SURVEY 8(a) pixel sum


def test_umax_and_gaussian_kernel(orc):
    assert list(orc.umax(15)) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert list(orc.gaussian_kernel7_q8()) == [18, 34, 49, 55, 49, 34, 18]


# ---------------------------------------------------------------- FAST
def test_fast_bright_ring_score(orc):
    img = np.full((16, 16), 200, np.uint8)
    img[8, 8] = 100                                   # every ring pixel is 100 brighter than the centre
    xs, ys, sc, smap = orc.fast_detect(img, 20)
    assert smap[8, 8] == 99                           # largest threshold that still passes: 100 - 1
    assert (list(xs), list(ys), list(sc)) == ([8], [8], [99])


def test_fast_needs_nine_contiguous(orc):
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    for n, expect in ((8, 0), (9, 1)):
        img = np.full((16, 16), 100, np.uint8)
        for dx, dy in ring[:n]:
            img[8 + dy, 8 + dx] = 160
        xs, ys, sc, smap = orc.fast_detect(img, 20)
        assert (smap[8, 8] > 0) == bool(expect), n
        if expect:
            assert smap[8, 8] == 59                   # darker-centre arc of exactly 9: min(|d|) - 1


def test_fast_nms_strict_maximum(orc):
    img = np.full((24, 24), 200, np.uint8)
    img[10, 10] = 100
    img[10, 11] = 100                                 # two adjacent equally strong corners: neither is a strict max
    xs, ys, sc, smap = orc.fast_detect(img, 20)
    assert smap[10, 10] > 0 and smap[10, 11] > 0 and smap[10, 10] == smap[10, 11]
    assert len(xs) == 0


def test_fast_border_excluded(orc):
    img = np.full((16, 16), 200, np.uint8)
    img[2, 8] = 100                                   # closer than 3 px to the top: never evaluated
    assert len(orc.fast_detect(img, 20)[0]) == 0


# ---------------------------------------------------------------- resize / pyramid / blur
def test_resize_constant_and_monotone(orc):
    assert (orc.resize_linear(np.full((48, 60), 77, np.uint8), 50, 40) == 77).all()
    ramp = np.tile(np.arange(0, 240, 2, dtype=np.uint8), (30, 1))          # 120 wide
    out = orc.resize_linear(ramp, 100, 25)
    ref = ((np.arange(100) + 0.5) * 1.2 - 0.5) * 2
    assert np.abs(out[0].astype(float) - ref).max() <= 1.0                  # bilinear within one grey level
    assert (np.diff(out[0].astype(int)) >= 0).all()


def test_half_pyramid_is_box_mean(orc):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (64, 96), dtype=np.uint8)
    lv = orc.half_pyramid(img)
    a = img.astype(int)
    box = (a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2
    assert (lv[1] == box).all() and lv[4].shape == (4, 6)


def test_blur_fixed_point(orc):
    assert (orc.gaussian_blur7(np.full((20, 20), 100, np.uint8)) == 101).all()      # (100*257*257 + 2^15) >> 16 = 101: taps sum to 257
    assert (orc.gaussian_blur7(np.full((20, 20), 255, np.uint8)) == 255).all()      # 257 saturates
    imp = np.zeros((21, 21), np.uint8)
    imp[10, 10] = 255
    k = np.array([18, 34, 49, 55, 49, 34, 18])
    exp = (np.outer(k, k) * 255 + 32768) >> 16
    assert (orc.gaussian_blur7(imp)[7:14, 7:14] == exp).all()


# ---------------------------------------------------------------- float helpers
def test_fast_atan2_and_sincos(orc):
    rng = np.random.default_rng(1)
    for y, x in rng.normal(0, 100, (200, 2)):
        a = orc.fast_atan2(np.float32(y), np.float32(x))
        t = math.degrees(math.atan2(y, x)) % 360
        assert min(abs(a - t), 360 - abs(a - t)) < 0.3
    for x in np.linspace(0, 2 * math.pi, 1001):
        s, c = orc.sincos_det(float(x))
        assert abs(s - math.sin(x)) < 4e-16 and abs(c - math.cos(x)) < 4e-16


# ---------------------------------------------------------------- matcher
def test_knn_known_bit_patterns(vislam, orc):
    d1 = np.zeros((2, 32), np.uint8)
    d1[1, :] = 0xFF
    d2 = np.zeros((3, 32), np.uint8)
    d2[0, 0] = 0x0F                # 4 bits from row 0, 252 from row 1
    d2[1, :] = 0xFF                # 256 / 0
    d2[2, 0] = 0xF0                # 4 bits from row 0: tie with train 0 -> lower index first
    o12, o21 = orc.knn2_hamming(d1, d2)
    assert list(o12["trainIdx"][0]) == [0, 2] and list(o12["distance"][0]) == [4.0, 4.0]
    assert list(o12["trainIdx"][1]) == [1, 0] and list(o12["distance"][1]) == [0.0, 252.0]
    assert list(o21["trainIdx"][1]) == [1, 0] and list(o21["distance"][1]) == [0.0, 256.0]
    assert (o12["imgIdx"] == 0).all() and (o12["queryIdx"][:, 0] == [0, 1]).all()


def test_knn_single_train_row(orc):
    o12, o21 = orc.knn2_hamming(np.zeros((3, 32), np.uint8), np.zeros((1, 32), np.uint8))
    assert (o12["trainIdx"][:, 0] == 0).all() and (o12["trainIdx"][:, 1] == -1).all()


def _knn(vislam, best, d0, second, d1):
    o = np.zeros((len(best), 2), vislam.DMATCH_DTYPE)
    o["queryIdx"] = np.arange(len(best))[:, None]
    o["trainIdx"][:, 0], o["trainIdx"][:, 1] = best, second
    o["distance"][:, 0], o["distance"][:, 1] = d0, d1
    return o


def test_ratio_symmetry_and_modes(vislam, orc):
    KP = vislam.KEYPOINT_DTYPE
    k1, k2 = np.zeros(4, KP), np.zeros(4, KP)
    k1["x"], k1["y"] = [10, 20, 30, 40], [10, 10, 10, 10]
    p = P(vislam, w_size=70, h_size=70)
    # q0: passes ratio, mutual.  q1: fails ratio (81 > 0.8*100).  q2: passes, not mutual.  q3: passes, mutual,
    # but direction 2 fails ITS ratio (only matters in INTENDED mode).
    k12 = _knn(vislam, [0, 1, 3, 3], [8, 81, 10, 10], [1, 0, 0, 0], [10, 100, 100, 100])
    k21 = _knn(vislam, [0, 1, 0, 3], [8, 81, 50, 10], [1, 0, 1, 0], [100, 100, 100, 11])
    g, s = orc.good_matches(p, k1, k2, k12, k21)
    assert list(s["queryIdx"]) == [0, 3] and list(s["trainIdx"]) == [0, 3] and (s["imgIdx"] == -1).all()
    p.sym_mode = 1
    g2, s2 = orc.good_matches(p, k1, k2, k12, k21)
    assert list(s2["queryIdx"]) == [0]
    # exact ratio boundary: d0 == (double)0.8f * d1 is NOT removed only if d0 <= product; 0.8f > 0.8
    k12b = _knn(vislam, [0], [80], [1], [100])
    k21b = _knn(vislam, [0], [80], [1], [200])
    assert len(orc.good_matches(P(vislam, w_size=70, h_size=70), k1[:1], k2[:1], k12b, k21b)[1]) == 1


def test_grid_filter_cells_order_and_ties(vislam, orc):
    KP = vislam.KEYPOINT_DTYPE
    p = P(vislam, w_size=70, h_size=70, n_cells=49)          # 7x7 cells of 10x10
    xs = [5, 6, 15, 65, 5, 69.5]
    ys = [5, 6, 5, 5, 25, 69.5]
    k1, k2 = np.zeros(6, KP), np.zeros(6, KP)
    k1["x"], k1["y"] = xs, ys
    n = 6
    dist = [30, 20, 20, 7, 9, 11]
    k12 = _knn(vislam, list(range(n)), dist, [(i + 1) % n for i in range(n)], [200] * n)
    k21 = _knn(vislam, list(range(n)), dist, [(i + 1) % n for i in range(n)], [200] * n)
    g, s = orc.good_matches(p, k1, k2, k12, k21)
    assert len(s) == 6
    # band 0: cell 0 holds q0 (d30) and q1 (d20) -> q1; cell 1 -> q2; cell 6 -> q3.  band 2: q4.  band 6: q5.
    assert list(g["queryIdx"]) == [1, 2, 3, 4, 5]
    # tie inside a cell: the first in y-sorted order wins (strict <)
    k12["distance"][0, 0] = 20
    k21["distance"][0, 0] = 20
    g, s = orc.good_matches(p, k1, k2, k12, k21)
    assert list(g["queryIdx"])[0] == 0
    # x beyond w_size: column index clamped to root-1 (SPEC, reference would write out of bounds)
    k1["x"][3] = 500.0
    g, s = orc.good_matches(p, k1, k2, k12, k21)
    assert 3 in list(g["queryIdx"])
    # empty input
    g, s = orc.good_matches(p, k1[:0], k2[:0], k12[:0], k21[:0])
    assert len(g) == 0 and len(s) == 0


# ---------------------------------------------------------------- pose
def _two_view(n, seed):
    import test_pose_gpu
    return test_pose_gpu.two_view(n, seed)


def test_five_point_recovers_true_E_and_satisfies_constraints(vislam, orc):
    x1, x2, R, t = _two_view(5, 42)
    f, cx, cy = 458.654, 367.215, 248.375
    q1 = (x1.astype(np.float64) - [cx, cy]) / f
    q2 = (x2.astype(np.float64) - [cx, cy]) / f
    Es = orc.five_point(q1, q2)
    assert 1 <= len(Es) <= 10
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Et = tx @ R
    Et /= np.linalg.norm(Et)
    best = min(min(np.abs(E - Et).max(), np.abs(E + Et).max()) for E in Es)
    assert best < 1e-5                                          # float32 pixel coordinates limit this
    h1 = np.c_[q1, np.ones(5)]
    h2 = np.c_[q2, np.ones(5)]
    for E in Es:
        assert abs(np.linalg.norm(E) - 1) < 1e-12
        assert np.abs(np.einsum("ni,ij,nj->n", h2, E, h1)).max() < 1e-10
        assert abs(np.linalg.det(E)) < 1e-10
        assert np.abs(2 * E @ E.T @ E - np.trace(E @ E.T) * E).max() < 1e-9


def test_ransac_sample_stream_properties(orc):
    s = orc.ransac_samples(0xFFFFFFFFFFFFFFFF, 49, 1000)
    assert s.min() >= 0 and s.max() < 49
    assert all(len(set(r)) == 5 for r in s.tolist())
    assert (s == orc.ransac_samples(0xFFFFFFFFFFFFFFFF, 49, 1000)).all()
    assert not (s[:10] == orc.ransac_samples(12345, 49, 10)).all()


def test_ransac_and_recover_pose_on_exact_data(vislam, orc):
    p = P(vislam)
    p.fy = p.fx
    x1, x2, R, t = _two_view(60, 7)
    E, mask, ninl, iters = orc.essential_ransac(p, x1, x2)
    assert ninl == 60 and mask.all() and 1 <= iters <= 3
    Rr, tr, ng = orc.recover_pose(p, E, x1, x2)
    assert ng == 60 and np.abs(Rr - R).max() < 1e-4 and np.abs(tr - t).max() < 1e-3
    assert abs(np.linalg.det(Rr) - 1) < 1e-9
    # degenerate sizes
    E4, m4, n4, i4 = orc.essential_ransac(p, x1[:4], x2[:4])
    assert n4 == 0 and (E4 == 0).all()
    E5, m5, n5, i5 = orc.essential_ransac(p, x1[:5], x2[:5])
    assert n5 == 5 and i5 == 1


def test_f2f_ransac_pure_translation(vislam, orc):
    p = P(vislam)
    KP = vislam.KEYPOINT_DTYPE
    rng = np.random.default_rng(2)
    n = 30
    X = np.c_[rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(4, 9, n)]
    tv = np.array([0.3, -0.1, 0.05])
    a, b = np.zeros(n, KP), np.zeros(n, KP)
    a["x"], a["y"] = p.fx * X[:, 0] / X[:, 2] + p.cx, p.fy * X[:, 1] / X[:, 2] + p.cy
    X2 = X + tv
    b["x"], b["y"] = p.fx * X2[:, 0] / X2[:, 2] + p.cx, p.fy * X2[:, 1] / X2[:, 2] + p.cy
    idx = rng.integers(0, n - 1, (200, 2)).astype(np.int32)
    out, cm = orc.f2f_ransac(p, a, b, np.eye(3, dtype=np.float32), idx, 1.0)
    d = out / np.linalg.norm(out)
    assert abs(abs(d @ (tv / np.linalg.norm(tv))) - 1) < 1e-3 and cm == n
    z, c0 = orc.f2f_ransac(p, a[:1], b[:1], np.eye(3, dtype=np.float32), idx[:0], 1.0)
    assert (z == 0).all()


# ---------------------------------------------------------------- Camera::computeGradient / patch builders
def test_scharr_gradient_known_values(orc):
    # constant image: no gradient anywhere (BORDER_REFLECT_101 keeps it constant)
    gx, gy, g = orc.scharr_gradient(np.full((12, 20), 99, np.uint8))
    assert not gx.any() and not gy.any() and not g.any()
    # horizontal ramp I = 2x: dx = (3+10+3) * (I(x+1) - I(x-1)) * scale = 16 * 4 * 3 = 192 in the interior, dy = 0;
    # reflect-101 makes the two border columns see I(x+1) - I(x-1) = 0
    ramp = np.tile((2 * np.arange(20)).astype(np.uint8), (12, 1))
    gx, gy, g = orc.scharr_gradient(ramp, 3)
    assert (gx[:, 1:-1] == 192).all() and (gx[:, 0] == 0).all() and (gx[:, -1] == 0).all()
    assert not gy.any()
    assert (g[:, 1:-1] == 96).all()                       # (192 + 0) / 2
    # vertical step: sign of dy (below minus above) and saturation of |.| before the blend
    img = np.zeros((10, 16), np.uint8); img[5:] = 255
    gx, gy, g = orc.scharr_gradient(img, 3)
    assert (gy[4] == 16 * 255 * 3).all() and (gy[5] == 16 * 255 * 3).all() and (gy[3] == 0).all()
    assert (g[4] == 128).all()                            # (255 + 0) / 2 = 127.5 -> round half to even = 128


def test_gradient_blend_rounds_half_to_even(orc):
    # |dx| = 1*scale... build values whose half sums hit .5 with even and odd floors: use scale 1 and tiny steps
    img = np.zeros((8, 16), np.uint8)
    img[:, 8:] = 1                                         # dx = 16 at the step columns, 0 elsewhere -> g = 8
    gx, gy, g = orc.scharr_gradient(img, 1)
    assert gx[4, 7] == 16 and gx[4, 8] == 16 and g[4, 7] == 8
    img = np.zeros((8, 16), np.uint8); img[3, 5] = 1       # a single bright pixel: |dx| = 3 / 10, |dy| = 3 / 10 around it
    gx, gy, g = orc.scharr_gradient(img, 1)
    assert gx[3, 4] == 10 and gy[3, 4] == 0 and g[3, 4] == 5
    assert abs(gx[2, 4]) == 3 and abs(gy[2, 4]) == 3 and g[2, 4] == 3
    assert abs(gx[2, 5]) == 0 and abs(gy[2, 5]) == 10 and g[2, 5] == 5
    # 3 and 0 -> 1.5 -> 2 (even); 3 and 10 -> 6.5 -> 6 (even)
    img = np.zeros((8, 16), np.uint8); img[3, 5] = 1; img[3, 9] = 1
    gx, gy, g = orc.scharr_gradient(img, 1)
    assert g[2, 4] == 3


def test_patch_points_reference_quirks(orc):
    from vislam import KEYPOINT_DTYPE
    good = np.zeros(1, KEYPOINT_DTYPE); good["x"] = 400.0; good["y"] = 300.0
    # `patch_size - 1 / 2` is integer arithmetic: start_point == patch_size (5, 3, 2, 5, 5) -> (2*sp+1)^2 points
    for l, sp in enumerate((5, 3, 2, 5, 5)):
        pts = orc.patch_points(good, 752, 480, l)
        assert len(pts) == (2 * sp + 1) ** 2, l
        x = (400.0 + 0.5) / 2 ** l - 0.5
        assert pts[0, 0] == int(x - sp) and pts[0, 2] == 1.0 and pts[0, 3] == 1.0
        assert (pts[1, 0] == pts[0, 0]) and (pts[1, 1] == pts[0, 1] + 1)          # inner loop runs over y
    # points with i <= 0 or j <= 0 are dropped (strict comparisons in the reference)
    good["x"] = 1.0; good["y"] = 1.0
    pts = orc.patch_points(good, 752, 480, 0)
    assert (pts[:, 0] > 0).all() and (pts[:, 1] > 0).all() and len(pts) == 6 * 6
    # at most 200 keypoints are used
    many = np.zeros(300, KEYPOINT_DTYPE); many["x"] = 300; many["y"] = 200
    assert len(orc.debug_points(many, 2)) == 200
    assert orc.debug_points(many, 2)[0, 0] == np.float32((300 + 0.5) * 0.25 - 0.5)


def test_pipeline_stream_mt_equals_sequential(vislam, orc):
    """the frame-parallel CPU baseline runs exactly the per-frame pipeline (bench.py cpu_baseline_multicore)"""
    p = vislam.default_params(); p.fy = p.fx
    p.nfeatures, p.w_size, p.h_size = 300, 320, 240
    cv = vislam.synth_canvas(1024, 9)
    fr = np.stack([vislam.synth_frame(cv, t, 320, 240, 9) for t in range(6)])
    prev, ref = None, []
    for t in range(6):
        k, d, r = orc.pipeline_frame(p, fr[t], prev)
        prev = (k, d)
        ref.append((r.n_kp, r.n_sym, r.n_good, r.n_inliers, r.n_pose_good, r.iters_run, tuple(r.E)))
    for th in (1, 3):
        sec, res = orc.pipeline_stream_mt(p, fr, th)
        assert sec > 0
        assert [(r.n_kp, r.n_sym, r.n_good, r.n_inliers, r.n_pose_good, r.iters_run, tuple(r.E)) for r in res] == ref


def test_half_pyramid_sizes_follow_cv_resize(orc):
    """Camera::Update, src/Camera.cpp:68-70: resize(prev, next, Size(), 0.5, 0.5) -> dsize = cvRound(size * 0.5), round half to even"""
    lw, lh = orc.half_pyramid_dims(1920, 1080)
    assert lw == [1920, 960, 480, 240, 120] and lh == [1080, 540, 270, 135, 68]      # 135 * 0.5 = 67.5 -> 68 (the reference's h_size >> 4 says 67)
    assert orc.half_pyramid_dims(137, 133)[0][:2] == [137, 68] and orc.half_pyramid_dims(137, 133)[1][:2] == [133, 66]   # 68.5 -> 68, 66.5 -> 66
    assert orc.half_pyramid_dims(752, 480) == ([752, 376, 188, 94, 47], [480, 240, 120, 60, 30])


def test_half_pyramid_partial_blocks(orc):
    """area-fast resize: complete 2x2 blocks -> (a+b+c+d+2)>>2; the last column / row of a level that is one larger than half of an odd
    size averages the pixels that exist with saturate_cast<uchar>((float)sum / count) = round half to EVEN (hand-made example)"""
    img = np.zeros((18, 19), np.uint8)              # 19 -> 10 (9.5 -> 10): last column from source column 18 alone; 18 -> 9 rows, all complete
    img[:, 18] = np.arange(18) * 3
    img[0, 0:2] = (1, 2); img[1, 0:2] = (2, 2)      # (1+2+2+2+2)>>2 = 2  (7/4 = 1.75 rounds up; 9 >> 2 = 2)
    lv = orc.half_pyramid(img)
    assert lv[1].shape == (9, 10) and lv[1][0, 0] == 2
    # last column: rows 2y, 2y+1 of source column 18: (3*2y + 3*(2y+1)) / 2 = 6y + 1.5 -> round half to even: 6y + 2
    assert [int(v) for v in lv[1][:, 9]] == [6 * y + 2 for y in range(9)]
    # a level with an odd number of ROWS whose half rounds up: 22 rows -> 11 -> 6 (5.5 -> 6): last row of level 2 from one source row
    img2 = np.tile(np.arange(22, dtype=np.uint8)[:, None] * 10, (1, 32))
    lv2 = orc.half_pyramid(img2)
    assert lv2[1].shape == (11, 16) and lv2[2].shape == (6, 8)
    assert (lv2[2][5] == lv2[1][10, 0]).all()       # the mean of two equal pixels


def test_pipeline_frame_on_a_size_that_does_not_halve_exactly(orc, vislam, canvas):
    """orc_pipeline_frame builds Camera::Update's half pyramid every frame: its level buffers must have cv::resize's rounded sizes
    (375 -> 188 rows, not 375 >> 1 = 187: the first form overflowed its buffers at 500 x 375; found by tools/stress_batch.py, runs
    under ASan / UBSan in tests/test_oracle_asan.py like every test of this file)."""
    p = vislam.default_params()
    p.w_size, p.h_size, p.nfeatures, p.nlevels = 150, 110, 100, 3
    p.fy = p.fx
    prev = None
    for t in range(3):
        k, d, r = orc.pipeline_frame(p, vislam.synth_frame(canvas, 20 + t, 150, 110), prev)
        prev = (k, d)
        assert r.n_kp == len(k) > 20
    assert r.n_sym > 10
