"""CPU: independent re-derivations (numpy only) of whole oracle stages, and the numpy model of k_fast's byte-SWAR pretest.

These do NOT pin the oracle against OpenCV (nothing here can: OpenCV is absent and the reference ships no fixture -- parity
stays UNPINNED).  They shrink the surface where the oracle could be wrong about the *published
algorithms*: a second implementation written from the definition, not from the oracle's code, has to agree with it.
  (a) FAST-9/16: definitional score (largest threshold for which a 9-arc exists) + 3x3 NMS    vs orc_fast_detect
  (b) five-point: numpy.roots of the degree-10 polynomial                                     vs the bisection roots
  (c) cv::resize INTER_LINEAR: float64 bilinear interpolation at the same sample positions    vs orc_resize_linear (+-1 LSB)
  (d) k_fast's SWAR pretest (vi-slam_amd/csrc/detect.hip) replayed in numpy uint32 arithmetic: it never rejects a FAST corner
  (e) the root finder in the form the pose kernels run it (fixed levels on a zero-padded polynomial, compacted interval walk,
      guard-free bisection steps) replayed in Python floats                                   vs the sequential algorithm, bit for bit
  (f) BF-Hamming 2-NN from unpacked bits + stable sort                                         vs orc_knn2_hamming (exact, ties included)
  (g) Harris response and intensity-centroid angle of the oracle's keypoints in float64        vs the oracle (2e-5 relative / 0.35 degree)
  (h) the 8-bit Gaussian as exact int64 sums of the Q8 taps; rBRIEF descriptors re-derived from
      the blurred level (float32 rotation, round-half-even, LSB-first packing)                 vs the oracle (exact)
  (i) recoverPose from numpy SVDs (four candidates, DLT triangulation, cheirality vote)        vs orc_recover_pose (same winner / count, 1e-9)
  (j) the cyclic Jacobi iteration on A^T A and on D A^T A D, D = diag(1, 1, 1, -1) (t -> -t): V' = D V D bit for bit -- the
      identity k_pose_final uses to triangulate once for (R, t) and (R, -t)
  (k) Matcher::computeBestMatches (ratio, mutual best, y sort, 7 x 7 grid with float32 running bounds) from the reference's
      control flow, both symmetry modes                                                        vs orc_good_matches (exact)
  (l) VISystem::F2FRansac from the reference's statements (bearings, epipolar-plane normals, log10 inlier test, strictly
      larger count wins)                                                                       vs orc_f2f_ransac (same count, 1e-6)
  (m) the RANSAC loop of findEssentialMat in Python (cv::RNG stream, getSubset, float Sampson test, accept rule,
      RANSACUpdateNumIters; candidate models from the oracle's solver)                         vs orc_essential_ransac (iterations, mask, E exact)
"""
import numpy as np
import pytest

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _images(rng, n=96):
    yy, xx = np.mgrid[0:n, 0:n]
    noise = rng.integers(0, 256, (n, n)).astype(np.uint8)                                   # uniform noise: corners everywhere
    blocks = (rng.integers(0, 2, (n // 8, n // 8)) * 200 + 20).astype(np.uint8).repeat(8, 0).repeat(8, 1)
    blocks = np.clip(blocks.astype(int) + rng.integers(-6, 7, (n, n)), 0, 255).astype(np.uint8)   # checkerboard-like + noise
    ramp = np.clip(xx * 2 + yy + rng.integers(-30, 31, (n, n)), 0, 255).astype(np.uint8)     # gradient + strong noise
    extreme = (rng.integers(0, 2, (n, n)) * 255).astype(np.uint8)                            # 0 / 255 only: every wrap case of the SWAR bytes
    return {"noise": noise, "blocks": blocks, "ramp": ramp, "extreme": extreme}


def _definitional_fast(img, t):
    """score[y, x] = largest t' such that 9 contiguous ring pixels are all > c + t' or all < c - t' (0 if t' < t)"""
    h, w = img.shape
    c = img[3:h - 3, 3:w - 3].astype(np.int32)
    d = np.stack([img[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx].astype(np.int32) for dx, dy in RING]) - c[None]
    dd = np.concatenate([d, d[:8]])
    bright = np.max(np.stack([dd[k:k + 9].min(0) for k in range(16)]), 0)       # max over arcs of the smallest (ring - c)
    dark = np.max(np.stack([(-dd[k:k + 9]).min(0) for k in range(16)]), 0)
    s = np.maximum(bright, dark) - 1
    score = np.zeros((h, w), np.int32)
    score[3:h - 3, 3:w - 3] = np.where(s >= t, s, 0)
    return score


@pytest.mark.parametrize("t", [5, 20, 60])
def test_definitional_fast_matches_the_oracle(orc, t):
    rng = np.random.default_rng(100 + t)
    for name, img in _images(rng).items():
        score = _definitional_fast(img, t)
        xs, ys, sc, smap = orc.fast_detect(img, t)
        assert np.array_equal(smap.astype(np.int32), np.minimum(score, 255)), name
        pad = np.pad(score, 1)
        h, w = score.shape
        nb = np.stack([pad[1 + dy:1 + dy + h, 1 + dx:1 + dx + w] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dx, dy) != (0, 0)])
        keep = (score > 0) & (score[None] > nb).all(0)
        ky, kx = np.nonzero(keep)                                                # row-major, like cv::FAST emits
        assert np.array_equal(xs, kx) and np.array_equal(ys, ky), name
        assert np.array_equal(sc, score[ky, kx]), name


def test_degree10_roots_are_complete(orc):
    """numpy.roots (companion-matrix eigenvalues) vs the oracle's derivative-interlacing bisection on 200 random minimal
    samples: same number of real roots, same values; and every returned E satisfies the five epipolar constraints"""
    rng = np.random.default_rng(7)
    checked = 0
    for trial in range(200):
        # a random two-view geometry: 5 points in front of both cameras
        ang = rng.normal(0, 0.2, 3)
        K = np.array([[0, -ang[2], ang[1]], [ang[2], 0, -ang[0]], [-ang[1], ang[0], 0]])
        R = np.eye(3) + K + K @ K / 2
        U, _, Vt = np.linalg.svd(R); R = U @ Vt
        tvec = rng.normal(0, 1, 3); tvec /= np.linalg.norm(tvec)
        X = np.column_stack([rng.uniform(-1, 1, 5), rng.uniform(-1, 1, 5), rng.uniform(2, 6, 5)])
        X2 = X @ R.T + tvec
        q1, q2 = X[:, :2] / X[:, 2:], X2[:, :2] / X2[:, 2:]
        Es, poly, roots = orc.five_point_poly(q1, q2)
        if np.abs(poly).max() == 0:
            continue
        allr = np.roots(poly[::-1])
        scale = max(1.0, np.abs(allr).max())
        real = np.sort(allr[np.abs(allr.imag) < 1e-7 * scale].real)
        # a double root that numpy splits into a complex pair (or vice versa) is not a disagreement: skip near-degenerate cases
        gaps = np.abs(allr.imag[np.abs(allr.imag) >= 1e-7 * scale])
        if len(gaps) and gaps.min() < 1e-4 * scale:
            continue
        assert len(real) == len(roots), (trial, real, roots)
        assert np.allclose(real, roots, rtol=1e-6, atol=1e-8), (trial, real, roots)
        assert len(Es) == len(roots)
        h1 = np.column_stack([q1, np.ones(5)]); h2 = np.column_stack([q2, np.ones(5)])
        for E, z in zip(Es, roots):
            assert np.abs(np.einsum("ni,ij,nj->n", h2, E, h1)).max() < 1e-8      # always: E is in the null space by construction
            # the cubic constraints hold to rounding unless the root is nearly double (back-substitution through the cofactors of
            # B(z) is ill-conditioned there: one such sample, roots 1.07711 / 1.07768, reaches 5e-5)
            others = np.abs(allr - z); others = np.sort(others)[1] if len(allr) > 1 else 1.0
            if others < 1e-2 * max(1.0, abs(z)):
                continue
            assert abs(np.linalg.det(E)) < 1e-8
            assert np.abs(2 * E @ E.T @ E - np.trace(E @ E.T) * E).max() < 1e-7
        # the true essential matrix is among the solutions
        Etrue = np.array([[0, -tvec[2], tvec[1]], [tvec[2], 0, -tvec[0]], [-tvec[1], tvec[0], 0]]) @ R
        Etrue /= np.linalg.norm(Etrue)
        assert min(min(np.abs(E - Etrue).max(), np.abs(E + Etrue).max()) for E in Es) < 1e-6, trial
        checked += 1
    assert checked >= 150


@pytest.mark.parametrize("sw,sh,dw,dh", [(752, 480, 627, 400), (627, 400, 522, 333), (96, 64, 80, 53), (64, 64, 32, 32), (100, 37, 34, 13)])
def test_float64_bilinear_within_one_lsb(orc, sw, sh, dw, dh):
    """cv::resize INTER_LINEAR is a bilinear interpolation at ((d + 0.5) * scale - 0.5) with clamped borders, evaluated in
    11-bit fixed point: the float64 evaluation of the same formula may differ by rounding only"""
    rng = np.random.default_rng(sw * 7 + dw)
    src = rng.integers(0, 256, (sh, sw)).astype(np.uint8)
    got = orc.resize_linear(src, dw, dh).astype(np.float64)

    def axis(n_dst, n_src):
        f = (np.arange(n_dst) + 0.5) * (n_src / n_dst) - 0.5
        i0 = np.floor(f).astype(int)
        a = f - i0
        a = np.where(i0 < 0, 0.0, a); i0 = np.maximum(i0, 0)
        a = np.where(i0 >= n_src - 1, 0.0, a); i0 = np.minimum(i0, n_src - 1)
        return i0, np.minimum(i0 + 1, n_src - 1), a
    # rows: the reference implementation clamps the row INDICES but keeps the fraction (resizeGeneric_Invoker)
    fy = (np.arange(dh) + 0.5) * (sh / dh) - 0.5
    y0f = np.floor(fy).astype(int); ay = fy - y0f
    y0 = np.clip(y0f, 0, sh - 1); y1 = np.clip(y0f + 1, 0, sh - 1)
    x0, x1, ax = axis(dw, sw)
    s = src.astype(np.float64)
    top = s[y0][:, x0] * (1 - ax) + s[y0][:, x1] * ax
    bot = s[y1][:, x0] * (1 - ax) + s[y1][:, x1] * ax
    ref = top * (1 - ay)[:, None] + bot * ay[:, None]
    assert np.abs(got - ref).max() <= 1.0 + 1e-9


# ---- (d) the SWAR pretest of k_fast, formula for formula ---------------------------------------------------------------------
H = np.uint32(0x80808080); M7 = np.uint32(0x7F7F7F7F)


def _swar_unit_pass(C, N, S, E, W, t):
    """C, N, S, E, W: uint32 arrays, 4 pixels per word (centre and the ring pixels 3 px up / down / right / left).
    Returns the pass word (bit 7 of byte j = position j goes to cornerScore), exactly as detect.hip computes it."""
    K = (t + 1) >> 1
    assert 3 <= K <= 128
    kD = np.uint32((0x80808080 - (K - 1) * 0x01010101) & 0xFFFFFFFF)
    kB = np.uint32(((2 * K - 3) * 0x01010101) & 0xFFFFFFFF)
    one = np.uint32(1)
    cD = ((C >> one) & M7) + kD
    dn, ds = cD - ((N >> one) & M7), cD - ((S >> one) & M7)
    de, dw = cD - ((E >> one) & M7), cD - ((W >> one) & M7)
    dark = (dn | ds) & (de | dw)
    nbright = ((dn + kB) & (ds + kB)) | ((de + kB) & (dw + kB))
    return (dark | ~nbright) & H


def _pack4(a):
    a = a.astype(np.uint32)
    return a[..., 0] | (a[..., 1] << np.uint32(8)) | (a[..., 2] << np.uint32(16)) | (a[..., 3] << np.uint32(24))


@pytest.mark.parametrize("t", [5, 6, 7, 20, 21, 60, 120, 200, 255])
def test_swar_pretest_never_rejects_a_corner(t):
    rng = np.random.default_rng(t)
    with np.errstate(over="ignore"):
        for name, img in _images(rng, 128).items():
            h, w = img.shape
            score = _definitional_fast(img, t)
            wq = (w - 6) // 4 * 4
            cy, cx = np.mgrid[3:h - 3, 3:3 + wq:4]                               # first pixel of every 4-pixel unit
            def word(dy, dx):
                return _pack4(np.stack([img[cy + dy, cx + dx + j] for j in range(4)], -1))
            ps = _swar_unit_pass(word(0, 0), word(3, 0), word(-3, 0), word(0, 3), word(0, -3), t)
            for j in range(4):
                bit = (ps >> np.uint32(8 * j + 7)) & np.uint32(1)
                corner = score[cy, cx + j] > 0
                assert not (corner & (bit == 0)).any(), (name, t, j)
                # and it is a useful test, not "pass everything" (except on the 0/255 image, where every pixel is extreme)
            if name == "blocks" and 20 <= t <= 120:          # beyond that the wrap cases (|difference| > 128 - K) dominate: still safe, no longer selective
                allbits = sum(((ps >> np.uint32(8 * j + 7)) & np.uint32(1)).sum() for j in range(4))
                assert allbits < 0.6 * 4 * ps.size


def test_swar_byte_tests_exhaustive():
    """every (centre, ring) byte pair in one lane, neighbours at the wrap-prone extremes, all thresholds 5..255: a ring pixel
    that is darker than c - t (brighter than c + t) always sets the dark bit (clears the not-bright bit)"""
    c1, r1 = np.meshgrid(np.arange(256, dtype=np.uint32), np.arange(256, dtype=np.uint32), indexing="ij")
    one = np.uint32(1)
    ext = np.array([0, 1, 127, 128, 254, 255], np.uint32)
    with np.errstate(over="ignore"):
        for t in list(range(5, 256, 3)) + [20, 255]:
            K = (t + 1) >> 1
            kD = np.uint32((0x80808080 - (K - 1) * 0x01010101) & 0xFFFFFFFF)
            kB = np.uint32(((2 * K - 3) * 0x01010101) & 0xFFFFFFFF)
            for lane in range(4):
                for oc in ext:
                    for orr in ext:
                        C = np.zeros((256, 256), np.uint32); R = np.zeros((256, 256), np.uint32)
                        for L in range(4):
                            C |= (c1 if L == lane else oc) << np.uint32(8 * L)
                            R |= (r1 if L == lane else orr) << np.uint32(8 * L)
                        D = (((C >> one) & M7) + kD) - ((R >> one) & M7)
                        B = D + kB
                        dbit = (D >> np.uint32(8 * lane + 7)) & one
                        nb = (B >> np.uint32(8 * lane + 7)) & one
                        ci, ri = c1.astype(np.int64), r1.astype(np.int64)
                        assert not ((ri < ci - t) & (dbit == 0)).any(), (t, lane, oc, orr)
                        assert not ((ri > ci + t) & (nb == 1)).any(), (t, lane, oc, orr)


# ---------------------------------------------------------------------------------------------------------------------------
# (e) the root finder as the HIP kernels run it (vi-slam_amd/csrc/pose.hip: poly_prepare, roots_level_lane, BISECT_STEP) against the
#     sequential algorithm of oracle/pose.cpp real_roots(), both replayed in Python floats (IEEE double, no FMA): bit-identical roots.
#     What the kernels do differently: (1) a polynomial whose leading coefficients vanished runs through the FIXED ten levels with
#     those coefficients set to exactly 0, (2) the end-point signs of a level are evaluated once and a lane visits only its
#     sign-change intervals, (3) a bisection step is taken without the "mid point strictly inside" guard (a collapsed interval makes
#     the step a no-op) and the collapse is only looked at every eighth step.
def _horner(q, x):
    r = q[-1]
    for c in q[-2::-1]:
        r = r * x + c
    return r


def _roots_sequential(cin):
    """oracle/pose.cpp real_roots(), line by line"""
    deg = 10
    mx = max(abs(v) for v in cin)
    if mx == 0:
        return []
    c = [v / mx for v in cin]
    while deg > 0 and abs(c[deg]) < 1e-15:
        deg -= 1
    if deg == 0:
        return []
    B = max(abs(c[i] / c[deg]) for i in range(deg)) + 1.0
    prev = []
    for d in range(1, deg + 1):
        k = deg - d
        q = []
        for i in range(d + 1):
            f = 1.0
            for j in range(k):
                f *= float(i + k - j)
            q.append(c[i + k] * f)
        cur = []
        for j in range(len(prev) + 1):
            lo = -B if j == 0 else prev[j - 1]
            hi = B if j == len(prev) else prev[j]
            flo, fhi = _horner(q, lo), _horner(q, hi)
            if (flo < 0) == (fhi < 0):
                continue
            for _ in range(200 if d == deg else 40):
                m = 0.5 * (lo + hi)
                if m <= lo or m >= hi:
                    break
                if (_horner(q, m) < 0) == (flo < 0):
                    lo = m
                else:
                    hi = m
            cur.append(0.5 * (lo + hi))
        prev = cur
    return prev


def _roots_kernel_form(cin):
    """the kernels' formulation: ten fixed levels on the zero-padded polynomial, sign pass, compacted walk, guard-free steps"""
    mx = max(abs(v) for v in cin)
    if mx == 0:
        return []
    c = [v / mx for v in cin]                                 # (the solver stores the normalised polynomial in the record)
    deg = 10
    for i in range(10, 0, -1):                                # poly_prepare
        if deg == i and abs(c[i]) < 1e-15:
            deg = i - 1
    cd = c[deg]
    c = [c[i] if i <= deg else 0.0 for i in range(11)]
    B = 0.0
    for i in range(10):
        if i < deg:
            B = max(B, abs(c[i] / cd))
    B += 1.0
    prev = []
    for D in range(1, 11):
        K = 10 - D
        q = []
        for i in range(D + 1):
            f = 1.0
            for jj in range(K):
                f *= float(i + K - jj)
            q.append(c[i + K] * f)
        nprev = len(prev)
        ends = [-B] + prev + [B]                              # end points 0 .. nprev + 1
        neg = [_horner(q, x) < 0 for x in ends]
        todo = [j for j in range(nprev + 1) if neg[j] != neg[j + 1]]
        cur = []
        for j in todo:
            lo, hi = ends[j], ends[j + 1]
            for it in range(200 if D == 10 else 40):
                m = 0.5 * (lo + hi)
                fm = _horner(q, m)
                if (it & 7) == 7 and not (m > lo and m < hi):
                    break
                if (fm < 0) == neg[j]:
                    lo = m
                else:
                    hi = m
            cur.append(0.5 * (lo + hi))
        prev = cur
    return prev


def test_root_finder_kernel_formulation_is_the_sequential_algorithm(orc):
    rng = np.random.default_rng(77)
    polys = []
    for k in range(24):                                       # polynomials of the five-point solver itself (pins the Python replay to the C oracle)
        X = np.column_stack([rng.uniform(-1, 1, 5), rng.uniform(-1, 1, 5), rng.uniform(2, 6, 5)])
        w = rng.normal(0, 0.3, 3)
        th = np.linalg.norm(w)
        Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / max(th, 1e-12)
        R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        X2 = X @ R.T + rng.normal(0, 0.5, 3)
        _, poly, roots = orc.five_point_poly(X[:, :2] / X[:, 2:], X2[:, :2] / X2[:, 2:])
        if np.abs(poly).max() == 0:
            continue
        seq = _roots_sequential(list(poly))
        assert seq == list(roots), k                          # the replay IS the oracle's algorithm
        polys.append(list(poly))
    for k in range(24):                                       # synthetic: known real roots (clustered / spread), reduced degrees, tiny leading terms
        nreal = int(rng.integers(0, 11))
        r = np.sort(rng.uniform(-3, 3, nreal) * (10.0 ** rng.integers(-2, 2)))
        p = np.poly1d([1.0])
        for x in r:
            p *= np.poly1d([1.0, -x])
        for _ in range((10 - nreal) // 2):
            a, b = rng.uniform(-2, 2), rng.uniform(0.1, 2)
            p *= np.poly1d([1.0, -2 * a, a * a + b * b])
        c = list(p.coeffs[::-1]) + [0.0] * (10 - p.order)
        if k % 3 == 1:
            c[10] = 0.0 if k % 2 else 3e-17 * max(abs(v) for v in c)              # leading coefficient lost: degree 9 (or lower below)
        if k % 6 == 5:
            c[9] = 1e-18 * max(abs(v) for v in c)
        polys.append([float(v) for v in c])
    reduced = 0
    for c in polys:
        a, b = _roots_sequential(c), _roots_kernel_form(c)
        assert a == b, (c, a, b)                              # bit-identical (== on floats), same count, same order
        mx = max(abs(v) for v in c)
        reduced += abs(c[10] / mx) < 1e-15
    assert reduced >= 6                                       # the reduced-degree path was exercised


# ---------------------------------------------------------------------------------------------------------------------------
# (f) BF-Hamming 2-NN from unpacked bits with a stable sort; (g) Harris response and intensity-centroid angle of the oracle's
#     keypoints re-derived in float64 from the pyramid; (h) the fixed-point 7x7 Gaussian against a float64 one (+-1 LSB) and the
#     rBRIEF descriptors re-derived from the blurred level with the rotation / rounding / bit order written out here.
def test_knn2_hamming_from_unpacked_bits(orc):
    rng = np.random.default_rng(5)
    d1 = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    d2 = rng.integers(0, 256, (257, 32), dtype=np.uint8)
    d2[40] = d2[7]; d2[200] = d2[7]; d1[5] = d2[7]; d1[6] = d2[7] ^ np.uint8(1)          # exact ties and distance 0 / 1
    o12, o21 = orc.knn2_hamming(d1, d2)
    b1, b2 = np.unpackbits(d1, axis=1).astype(np.int16), np.unpackbits(d2, axis=1).astype(np.int16)
    dist = (b1[:, None, :] != b2[None, :, :]).sum(-1)
    for o, dm in ((o12, dist), (o21, dist.T)):
        order = np.argsort(dm, axis=1, kind="stable")[:, :2]                                # ties: lowest index first
        assert np.array_equal(o["trainIdx"], order)
        assert np.array_equal(o["distance"], np.take_along_axis(dm, order, 1).astype(np.float32))
        assert np.array_equal(o["queryIdx"], np.arange(len(dm))[:, None].repeat(2, 1))


def _pattern():
    import os
    import re
    txt = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vi-slam_amd", "csrc", "orb_pattern.inc")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    v = np.array([float(x.rstrip("f")) for x in re.findall(r"-?\d+(?:\.\d*)?f?", txt)], np.float32)
    assert v.size == 1024
    return v.reshape(256, 4)


def test_harris_angle_and_descriptors_rederived(vislam, orc):
    p = vislam.default_params()
    p.nfeatures, p.nlevels, p.w_size, p.h_size = 300, 4, 320, 240
    cv = vislam.synth_canvas(1024, 0xE0C00001)
    img = vislam.synth_frame(cv, 9, 320, 240)
    op = orc.Params()
    for f, _ in p._fields_:
        setattr(op, f, getattr(p, f))
    kps, desc = orc.orb_detect_compute(op, img)
    assert len(kps) > 150
    ws, hs, sc, _ = orc.level_geometry(op, 320, 240)
    levels = [img]
    for l in range(1, 4):
        levels.append(orc.resize_linear(levels[-1], int(ws[l]), int(hs[l])))
    # (h1) fixed-point blur vs float64 Gaussian (sigma 2, 7 taps, reflect-101 border)
    # cv::GaussianBlur on 8-bit images (OpenCV 3.2: createSeparableLinearFilter with an integer kernel, bits = 8): the float
    # taps are rounded to Q8 one by one -- [18 34 49 55 49 34 18], sum 257, i.e. a gain of (257/256)^2 -- both passes are exact
    # integer sums and the only rounding is the final (x + 2^15) >> 16.  Re-derived here in int64; the exact Gaussian is within
    # that gain + half a grey level.
    x = np.arange(7) - 3.0
    g = np.exp(-x * x / 8.0); g /= g.sum()
    gq = np.rint(g.astype(np.float32) * 256).astype(np.int64)
    assert gq.tolist() == [18, 34, 49, 55, 49, 34, 18]
    blurred = []
    for lv in levels:
        fb = orc.gaussian_blur7(lv)
        pad = np.pad(lv.astype(np.int64), 3, mode="reflect")
        hh = sum(gq[i] * pad[:, i:i + lv.shape[1]] for i in range(7))
        vv = sum(gq[i] * hh[i:i + lv.shape[0], :] for i in range(7))
        assert np.array_equal(fb.astype(np.int64), (vv + 32768) >> 16)
        padf = pad.astype(np.float64)
        he = sum(g[i] * padf[:, i:i + lv.shape[1]] for i in range(7))
        ve = sum(g[i] * he[i:i + lv.shape[0], :] for i in range(7))
        assert np.abs(fb - ve * (257.0 / 256.0) ** 2).max() <= 0.5 + 0.3                   # + the rounding of the individual taps
        blurred.append(fb)
    umax = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    pat = _pattern()
    bits_equal = 0
    for k, dsc in zip(kps, desc):
        l = int(k["octave"]); lv = levels[l].astype(np.float64); s = float(sc[l])
        xl, yl = int(round(float(k["x"]) / s)), int(round(float(k["y"]) / s))
        assert abs(xl * np.float32(s) - k["x"]) < 1e-3 and abs(yl * np.float32(s) - k["y"]) < 1e-3
        # (g1) Harris, block 7, k = 0.04
        a = b = c = 0.0
        for dy in range(-3, 4):
            for dx in range(-3, 4):
                q = lv[yl + dy - 1:yl + dy + 2, xl + dx - 1:xl + dx + 2]
                ix = (q[1, 2] - q[1, 0]) * 2 + (q[0, 2] - q[0, 0]) + (q[2, 2] - q[2, 0])
                iy = (q[2, 1] - q[0, 1]) * 2 + (q[2, 0] - q[0, 0]) + (q[2, 2] - q[0, 2])
                a += ix * ix; b += iy * iy; c += ix * iy
        r = (a * b - c * c - 0.04 * (a + b) ** 2) * (1.0 / (4 * 7 * 255.0)) ** 4
        assert abs(r - float(k["response"])) <= 2e-5 * max(abs(r), 1e-12), (r, k["response"])
        # (g2) intensity centroid over the radius-15 disc
        m10 = m01 = 0.0
        for v in range(-15, 16):
            u = np.arange(-umax[abs(v)], umax[abs(v)] + 1)
            row = lv[yl + v, xl + u]
            m10 += (u * row).sum(); m01 += v * row.sum()
        ang = np.degrees(np.arctan2(m01, m10)) % 360.0
        d = abs(ang - float(k["angle"])); d = min(d, 360.0 - d)
        assert d < 0.35, (ang, k["angle"])                                                   # fastAtan2 is a 0.3-degree polynomial
        # (h2) rBRIEF on the blurred level: rotate by the keypoint's angle in float32, round half to even, compare, pack LSB first
        ar = np.float32(k["angle"]) * np.float32(np.pi / 180.0)
        ca, sa = np.float32(np.cos(np.float64(ar))), np.float32(np.sin(np.float64(ar)))
        bl = blurred[l]
        def val(px, py):
            fx = np.float32(px * ca) - np.float32(py * sa)
            fy = np.float32(px * sa) + np.float32(py * ca)
            return bl[yl + np.rint(fy).astype(np.int64), xl + np.rint(fx).astype(np.int64)].astype(np.int32)
        t = (val(pat[:, 0], pat[:, 1]) < val(pat[:, 2], pat[:, 3])).astype(np.uint8)
        mine = np.packbits(t.reshape(32, 8)[:, ::-1], axis=1).ravel()
        bits_equal += int((np.unpackbits(mine) == np.unpackbits(dsc)).sum())
        assert np.array_equal(mine, dsc), (int(k["octave"]), xl, yl)
    assert bits_equal == 256 * len(kps)


# ---------------------------------------------------------------------------------------------------------------------------
# (i) recoverPose from its definition: numpy SVD of E, the four (R, t) candidates, DLT triangulation by numpy SVD, the cheirality
#     vote with the 50-unit distance bound -- against the oracle's Jacobi-SVD realisation (same winner, same count, R and t to 1e-9)
def test_recover_pose_from_numpy_svd(vislam, orc):
    rng = np.random.default_rng(21)
    p = vislam.default_params()
    p.fy = p.fx
    op = orc.Params()
    for f, _ in p._fields_:
        setattr(op, f, getattr(p, f))
    K = np.array([[p.fx, 0, p.cx], [0, p.fx, p.cy], [0, 0, 1.0]])
    for trial in range(6):
        n = 60
        X = np.column_stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(3, 9, n)])
        w = rng.normal(0, 0.15, 3)
        th = np.linalg.norm(w)
        Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / th
        R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        t = rng.normal(0, 1, 3); t /= np.linalg.norm(t)
        x1 = (K @ X.T).T; x1 = x1[:, :2] / x1[:, 2:]
        X2 = X @ R.T + t
        x2 = (K @ X2.T).T; x2 = x2[:, :2] / x2[:, 2:]
        if trial >= 3:                                                                       # some points behind / far away
            x2[:8] = rng.uniform(0, 480, (8, 2))
        x1 = x1.astype(np.float32); x2 = x2.astype(np.float32)
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        E = tx @ R
        Ro, to, no = orc.recover_pose(op, E, x1, x2)
        # --- from the definition
        q1 = (x1.astype(np.float64) - [p.cx, p.cy]) / p.fx
        q2 = (x2.astype(np.float64) - [p.cx, p.cy]) / p.fx
        U, _, Vt = np.linalg.svd(E)
        if np.linalg.det(U) < 0: U = -U
        if np.linalg.det(Vt) < 0: Vt = -Vt
        W = np.array([[0, 1.0, 0], [-1, 0, 0], [0, 0, 1]])
        cands = [(U @ W @ Vt, U[:, 2]), (U @ W.T @ Vt, U[:, 2]), (U @ W @ Vt, -U[:, 2]), (U @ W.T @ Vt, -U[:, 2])]
        P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
        good = []
        for Rc, tc in cands:
            P1 = np.hstack([Rc, tc[:, None]])
            cnt = 0
            for a, b in zip(q1, q2):
                A = np.array([a[0] * P0[2] - P0[0], a[1] * P0[2] - P0[1], b[0] * P1[2] - P1[0], b[1] * P1[2] - P1[1]])
                Q = np.linalg.svd(A)[2][-1]
                ok = Q[2] * Q[3] > 0
                Q = Q / Q[3]
                ok = ok and Q[2] < 50
                Q2 = P1 @ Q
                ok = ok and Q2[2] > 0 and Q2[2] < 50
                cnt += bool(ok)
            good.append(cnt)
        if good[0] >= max(good[1:]):
            sel = 0
        elif good[1] >= good[0] and good[1] >= good[2] and good[1] >= good[3]:
            sel = 1
        elif good[2] >= good[0] and good[2] >= good[1] and good[2] >= good[3]:
            sel = 2
        else:
            sel = 3
        assert no == good[sel], (good, no)
        assert np.abs(Ro - cands[sel][0]).max() < 1e-9 and np.abs(to - cands[sel][1]).max() < 1e-9
        if trial < 3:
            assert np.abs(Ro - R).max() < 1e-5 and np.abs(to - t).max() < 1e-5                # and it is the true motion


# ---------------------------------------------------------------------------------------------------------------------------
# (j) k_pose_final triangulates a correspondence once for [R | t] and derives the vote for [R | -t] from the same eigenvector:
#     negating t negates column 3 of the DLT matrix, A' = A D, and the cyclic Jacobi iteration (oracle/pose.cpp jacobi_eig) is
#     exactly equivariant under that sign change -- V' = D V D bit for bit -- unless some rotation with column 3 has theta == +-0.
def _jacobi(A):
    n = len(A)
    A = [row[:] for row in A]
    V = [[1.0 if i == j else 0.0 for j in range(n)] for i in range(n)]
    zero_theta = False
    for _ in range(30):
        off = 0.0
        for i in range(n):
            for j in range(i + 1, n):
                off += A[i][j] * A[i][j]
        if off < 1e-300:
            break
        for p in range(n):
            for q in range(p + 1, n):
                apq = A[p][q]
                if abs(apq) < 1e-300:
                    continue
                theta = (A[q][q] - A[p][p]) / (2.0 * apq)
                zero_theta |= (q == n - 1 and theta == 0.0)
                t = (1.0 if theta >= 0 else -1.0) / (abs(theta) + (theta * theta + 1.0) ** 0.5)
                c = 1.0 / (t * t + 1.0) ** 0.5
                s = t * c
                for k in range(n):
                    akp, akq = A[k][p], A[k][q]
                    A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq
                for k in range(n):
                    apk, aqk = A[p][k], A[q][k]
                    A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk
                for k in range(n):
                    vkp, vkq = V[k][p], V[k][q]
                    V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq
    return A, V, zero_theta


def test_jacobi_is_equivariant_under_negating_the_last_column():
    rng = np.random.default_rng(9)
    D = [1.0, 1.0, 1.0, -1.0]
    checked = 0
    for trial in range(300):
        x1, y1, x2, y2 = rng.uniform(-0.6, 0.6, 4)
        w = rng.normal(0, 0.3, 3); th = np.linalg.norm(w)
        Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / th
        R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        t = rng.normal(0, 1, 3); t /= np.linalg.norm(t)
        def ata(tt):
            P = [[float(R[r][c]) for c in range(3)] + [float(tt[r])] for r in range(3)]
            A = [[-1.0, 0.0, float(x1), 0.0], [0.0, -1.0, float(y1), 0.0],
                 [float(x2) * P[2][c] - P[0][c] for c in range(4)], [float(y2) * P[2][c] - P[1][c] for c in range(4)]]
            return [[sum(A[k][i] * A[k][j] for k in range(4)) for j in range(4)] for i in range(4)]
        Ap, Vp, z = _jacobi(ata(t))
        An, Vn, _ = _jacobi(ata(-t))
        if z:
            continue                                                                        # the kernel decomposes the second candidate itself then
        checked += 1
        for i in range(4):
            assert An[i][i] == Ap[i][i]
            for j in range(4):
                assert Vn[i][j] == D[i] * Vp[i][j] * D[j], (trial, i, j)
    assert checked >= 290


# ---------------------------------------------------------------------------------------------------------------------------
# (k) Matcher::computeBestMatches (src/Matcher.cpp:96-244, 309-352) written out again in Python from the reference's control
#     flow, with the behaviour DESIGN.md specifies where the reference has undefined behaviour (second direction effectively not
#     ratio-filtered, column index clamped, empty input -> empty output, stable y sort) -- against orc_good_matches on real keypoints
def _best_matches_py(p, kps1, knn12, knn21, intended):
    ratio = float(np.float32(p.ratio))                         # double nn_match_ratio = 0.8f
    def survives(row):                                         # nnFilter: two neighbours and not d0 > ratio * d1
        return row[1]["trainIdx"] >= 0 and not (float(row[0]["distance"]) > ratio * float(row[1]["distance"]))
    sym = []
    for q in range(len(knn12)):
        r1 = knn12[q]
        if r1[0]["trainIdx"] < 0 or not survives(r1):
            continue
        t = int(r1[0]["trainIdx"])
        r2 = knn21[t]                                          # the aux2 entry whose queryIdx is t
        if r2[0]["trainIdx"] < 0 or (intended and not survives(r2)):
            continue
        if int(r2[0]["trainIdx"]) == q:
            sym.append((q, t, np.float32(r1[0]["distance"])))
    order = sorted(range(len(sym)), key=lambda i: (np.float32(kps1[sym[i][0]]["y"]), i))       # sortIdx on pt.y, made stable
    srt = [sym[i] for i in order]
    root = int(np.floor(np.sqrt(p.n_cells)))
    winw = np.float32(p.w_size / np.floor(np.sqrt(p.n_cells)))
    winh = np.float32(p.h_size / np.floor(np.sqrt(p.n_cells)))
    good = []
    it = 0
    h_final = winh
    for _ in range(root):
        if it >= len(srt):
            break
        cell = [None] * root
        while it < len(srt) and np.float32(kps1[srt[it][0]]["y"]) <= h_final:
            w_final = winw
            i = 0
            while np.float32(kps1[srt[it][0]]["x"]) > w_final:
                w_final = np.float32(w_final + winw)
                i += 1
            i = min(i, root - 1)
            if cell[i] is None or srt[it][2] < cell[i][2]:
                cell[i] = srt[it]
            it += 1
        good += [c for c in cell if c is not None]
        h_final = np.float32(h_final + winh)
    return sym, good


@pytest.mark.parametrize("intended", [0, 1])
def test_match_filters_rederived(vislam, orc, intended):
    p = vislam.default_params()
    p.nfeatures, p.nlevels, p.w_size, p.h_size, p.sym_mode = 600, 8, 640, 480, intended
    op = orc.Params()
    for f, _ in p._fields_:
        setattr(op, f, getattr(p, f))
    cv = vislam.synth_canvas(2048, 0xE0C00001)
    frames = [vislam.synth_frame(cv, t, 640, 480) for t in (0, 3, 11)]
    det = [orc.orb_detect_compute(op, f) for f in frames]
    total = 0
    for (k1, d1), (k2, d2) in zip(det[:-1], det[1:]):
        knn12, knn21 = orc.knn2_hamming(d1, d2)
        good, sym = orc.good_matches(op, k1, k2, knn12, knn21)
        psym, pgood = _best_matches_py(p, k1, knn12, knn21, intended)
        assert [(int(m["queryIdx"]), int(m["trainIdx"]), float(m["distance"])) for m in sym] == [(q, t, float(d)) for q, t, d in psym]
        assert [(int(m["queryIdx"]), int(m["trainIdx"]), float(m["distance"])) for m in good] == [(q, t, float(d)) for q, t, d in pgood]
        assert 10 <= len(pgood) <= 49
        total += len(pgood)
    assert total > 40


# ---------------------------------------------------------------------------------------------------------------------------
# (l) VISystem::F2FRansac (src/VISystem.cpp:612-769) written out again with numpy from the reference's statements: unit bearings
#     in double from float pixel coordinates and float intrinsics, normal_i = v1 x (R v2), d = normalise(n_a x n_b) for the given
#     sample pairs, count of -1000 / log10(|d . n_i|) < threshold, strictly larger count wins, result = scale * (float)d
def test_f2f_ransac_rederived(vislam, orc):
    p = vislam.default_params()
    op = orc.Params()
    for f, _ in p._fields_:
        setattr(op, f, getattr(p, f))
    rng = np.random.default_rng(31)
    KP = vislam.KEYPOINT_DTYPE
    for trial, m in enumerate((40, 12, 3, 2)):
        w = rng.normal(0, 0.05, 3); th = np.linalg.norm(w)
        Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / th
        R = (np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx).astype(np.float32)
        X = np.column_stack([rng.uniform(-2, 2, m), rng.uniform(-1.5, 1.5, m), rng.uniform(3, 9, m)])
        t = rng.normal(0, 0.3, 3)
        a = np.zeros(m, KP); b = np.zeros(m, KP)
        X2 = (X - t) @ R.astype(np.float64)                                                  # frame 2 sees R^T (X - t)
        a["x"] = p.fx * X[:, 0] / X[:, 2] + p.cx; a["y"] = p.fy * X[:, 1] / X[:, 2] + p.cy
        b["x"] = p.fx * X2[:, 0] / X2[:, 2] + p.cx + rng.normal(0, 0.3, m); b["y"] = p.fy * X2[:, 1] / X2[:, 2] + p.cy + rng.normal(0, 0.3, m)
        idx = rng.integers(0, max(m - 1, 1), (1000, 2)).astype(np.int32)
        scale = 0.37
        got, cnt = orc.f2f_ransac(op, a, b, R, idx, scale)
        fx, fy, cx, cy = (np.float32(v) for v in (p.fx, p.fy, p.cx, p.cy))
        def bearing(k):
            v = np.array([np.float64((k["x"] - cx) / fx), np.float64((k["y"] - cy) / fy), 1.0])     # float arithmetic, then double
            return v / np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
        normals = [np.cross(bearing(a[i]), R.astype(np.float64) @ bearing(b[i])) for i in range(m)]
        best, cmax = np.zeros(3, np.float32), 0
        for i1, i2 in idx:
            d = np.cross(normals[i1], normals[i2])
            if not d.any():
                continue
            d = d / np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
            c = 0
            for n in normals:
                with np.errstate(divide="ignore"):
                    c += bool(-1000.0 / np.log10(abs(float(d @ n))) < p.f2f_threshold)
            if c > cmax:
                cmax, best = c, d.astype(np.float32)
        assert cnt == cmax, (trial, cnt, cmax)
        assert np.abs(got - np.float32(scale) * best).max() <= 1e-6, (trial, got, best)


# ---------------------------------------------------------------------------------------------------------------------------
# (m) the RANSAC loop of findEssentialMat (RANSACPointSetRegistrator::run) written out again in Python: cv::RNG's multiply-with-
#     carry stream, getSubset (five distinct indices, redraw on duplicates), float Sampson error against the float threshold,
#     "strictly more than max(best, 4) inliers" accept rule, RANSACUpdateNumIters -- with the oracle's minimal solver for the
#     candidate models -- against orc_essential_ransac: same iteration count, same mask, same E
def test_ransac_loop_rederived(vislam, orc):
    import math
    p = vislam.default_params()
    p.fy = p.fx
    op = orc.Params()
    for f, _ in p._fields_:
        setattr(op, f, getattr(p, f))
    rng = np.random.default_rng(4)
    for trial, (n, nout, noise) in enumerate(((49, 0, 0.0), (60, 18, 0.4), (25, 10, 0.8), (7, 1, 0.2))):
        X = np.column_stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(3, 9, n)])
        w = rng.normal(0, 0.1, 3); th = np.linalg.norm(w)
        Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / th
        R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        t = rng.normal(0, 0.5, 3)
        X2 = X @ R.T + t
        x1 = np.column_stack([p.fx * X[:, 0] / X[:, 2] + p.cx, p.fx * X[:, 1] / X[:, 2] + p.cy]) + rng.normal(0, noise, (n, 2))
        x2 = np.column_stack([p.fx * X2[:, 0] / X2[:, 2] + p.cx, p.fx * X2[:, 1] / X2[:, 2] + p.cy]) + rng.normal(0, noise, (n, 2))
        x2[:nout] = rng.uniform(0, 480, (nout, 2))
        x1 = x1.astype(np.float32); x2 = x2.astype(np.float32)
        oE, omask, oninl, oiters = orc.essential_ransac(op, x1, x2)
        # --- the loop, from its definition
        q1 = (x1.astype(np.float64) - [p.cx, p.cy]) * (1.0 / p.fx)
        q2 = (x2.astype(np.float64) - [p.cx, p.cy]) * (1.0 / p.fx)
        thr = np.float32((p.ransac_threshold / p.fx) ** 2)
        state = p.ransac_seed if p.ransac_seed else 0xFFFFFFFF
        def nxt():
            nonlocal state
            state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & 0xFFFFFFFFFFFFFFFF
            return state & 0xFFFFFFFF
        niters, best_good, best_mask, bestE, it = max(p.ransac_max_iters, 1), 0, np.zeros(n, np.uint8), np.zeros((3, 3)), 0
        while it < niters:
            idx = []
            while len(idx) < 5:
                v = nxt() % n
                if v not in idx:
                    idx.append(v)
            for E in orc.five_point(q1[idx], q2[idx]):
                Ex = q1 @ E[:, :2].T + E[:, 2]                    # rows of E applied to (x1, y1, 1)
                Et = q2 @ E[:2, :] + E[2, :]                      # E^T applied to (x2, y2, 1)
                num = ((q2[:, 0] * Ex[:, 0] + q2[:, 1] * Ex[:, 1]) + Ex[:, 2]) ** 2
                den = ((Ex[:, 0] ** 2 + Ex[:, 1] ** 2) + Et[:, 0] ** 2) + Et[:, 1] ** 2
                mask = ((num / den).astype(np.float32) <= thr).astype(np.uint8)
                good = int(mask.sum())
                if good > max(best_good, 4):
                    best_good, best_mask, bestE = good, mask, E
                    if p.ransac_adaptive:
                        ep = min(max((n - good) / n, 0.0), 1.0)
                        num_l = max(1.0 - p.ransac_prob, 2.2250738585072014e-308)
                        den_l = 1.0 - (1.0 - ep) ** 5
                        if den_l < 2.2250738585072014e-308:
                            niters = 0
                        else:
                            a, b = math.log(num_l), math.log(den_l)
                            niters = niters if (b >= 0 or -a >= niters * (-b)) else int(np.rint(a / b))
            it += 1
        assert (oninl, oiters) == (best_good, it), (trial, oninl, oiters, best_good, it)
        assert np.array_equal(omask, best_mask)
        assert np.abs(oE - bestE).max() == 0


def test_half_pyramid_against_a_numpy_restatement_at_odd_sizes(orc):
    """resizeAreaFast with scale 2 on sizes that do not halve exactly (1080p: 135 -> 68 rows), written out in numpy: complete blocks
    (sum + 2) >> 2, partial blocks np.rint(sum / count) (numpy's rint is round-half-even like cvRound)"""
    rng = np.random.default_rng(77)
    def half_dim(n):
        k = n >> 1
        return k if n % 2 == 0 else (k if k % 2 == 0 else k + 1)
    for h, w in ((135, 137), (1080 // 8, 1920 // 8), (33, 35), (61, 47), (270, 150)):
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        got = orc.half_pyramid(img)
        prev = img.astype(np.int64)
        for l in range(1, 5):
            ph, pw = prev.shape
            nh, nw = half_dim(ph), half_dim(pw)
            pad = np.zeros((2 * nh, 2 * nw), np.int64); cnt = np.zeros((2 * nh, 2 * nw), np.int64)
            pad[:min(ph, 2 * nh), :min(pw, 2 * nw)] = prev[:2 * nh, :2 * nw]; cnt[:min(ph, 2 * nh), :min(pw, 2 * nw)] = 1
            ssum = pad.reshape(nh, 2, nw, 2).sum((1, 3)); c = cnt.reshape(nh, 2, nw, 2).sum((1, 3))
            cur = np.where(c == 4, (ssum + 2) >> 2, np.rint(ssum / np.maximum(c, 1)).astype(np.int64))
            assert got[l].shape == (nh, nw), (h, w, l)
            assert np.array_equal(got[l], cur.astype(np.uint8)), (h, w, l)
            prev = cur


def test_single_precision_sampson_error_radii_hold():
    """vi-slam_amd/csrc/pose.hip decides the Sampson test in single precision inside rigorous error radii (|s32 - s| < 7 u n1 n2,
    |den32 - den| < 12 u (n1^2 + n2^2) for ||E||_F = 1, u = 2^-24, n = |(x, y, 1)|; the kernel uses 8 u and 16 u) and in the oracle's double
    sequence otherwise.  This replays the kernel's operation order in numpy float32 (each fmaf as an exactly-rounded a*b + c: the product
    of two floats is exact in float64, the double rounding is at most one float ulp of slack inside the margins) on two million random
    (model, point) pairs -- unit-norm E, coordinates up to |4| like the image corners of a short focal length -- against float64, and
    checks that no pair ever leaves the radii the kernel assumes, and that the "certainly in / certainly out" decisions
    (r = fma(-tm, den32, s32^2) against band = kd den32 + 2 es' |s32| + c, see the kernel's comment) never contradict the
    double-precision rule -- on random pairs and on pairs constructed to sit within a few 10^-7 (relative) of the threshold."""
    rng = np.random.default_rng(2024)
    f32 = np.float32
    u = 2.0 ** -24

    def fma(a, b, c):
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)

    def replay(E, P):
        """single-precision s (signed), den in the kernel's operation order; the double values; the kernel's radii"""
        n = len(E)
        e = [E[:, k].astype(f32) for k in range(9)]
        x1, y1, x2, y2 = (P[:, k].astype(f32) for k in range(4))
        ex0 = fma(e[0], x1, fma(e[1], y1, e[2])); ex1 = fma(e[3], x1, fma(e[4], y1, e[5])); ex2 = fma(e[6], x1, fma(e[7], y1, e[8]))
        et0 = fma(e[0], x2, fma(e[3], y2, e[6])); et1 = fma(e[1], x2, fma(e[4], y2, e[7]))
        s32 = fma(x2, ex0, fma(y2, ex1, ex2))
        den32 = fma(ex0, ex0, fma(ex1, ex1, fma(et0, et0, (et1.astype(np.float64) * et1.astype(np.float64)).astype(f32))))
        X1, Y1, X2, Y2 = P[:, 0], P[:, 1], P[:, 2], P[:, 3]
        Ex0 = E[:, 0] * X1 + E[:, 1] * Y1 + E[:, 2]; Ex1 = E[:, 3] * X1 + E[:, 4] * Y1 + E[:, 5]; Ex2 = E[:, 6] * X1 + E[:, 7] * Y1 + E[:, 8]
        Et0 = E[:, 0] * X2 + E[:, 3] * Y2 + E[:, 6]; Et1 = E[:, 1] * X2 + E[:, 4] * Y2 + E[:, 7]
        s = X2 * Ex0 + Y2 * Ex1 + Ex2; den = Ex0 ** 2 + Ex1 ** 2 + Et0 ** 2 + Et1 ** 2
        one = np.ones(n, f32)
        n1 = (np.sqrt(fma(x1, x1, fma(y1, y1, one)).astype(np.float64)).astype(f32) * f32(1 + 2.0 ** -20)).astype(f32)
        n2 = (np.sqrt(fma(x2, x2, fma(y2, y2, one)).astype(np.float64)).astype(f32) * f32(1 + 2.0 ** -20)).astype(f32)
        es = (f32(8 * u) * n1 * n2).astype(f32); ed = (f32(16 * u) * (n1 * n1 + n2 * n2).astype(f32)).astype(f32)
        return s32, den32, s, den, es, ed

    def decide(s32, den32, es, ed, thr):
        n = len(s32)
        thr2 = f32(thr * thr)
        tmid = 0.5 * (float(thr2) + float(np.nextafter(thr2, f32(np.inf))))
        tm, kd, thi16 = f32(tmid), f32(tmid * 2.0 ** -14 * (1 + 2.0 ** -23)), f32(tmid * (1 + 2.0 ** -16) * (1 + 2.0 ** -23))
        es2 = (f32(2) * es * f32(1 + 2.0 ** -10)).astype(f32); c = ((es * es + thi16 * ed) * f32(1 + 2.0 ** -10)).astype(f32)
        r = fma(np.full(n, -tm, f32), den32, (s32.astype(np.float64) * s32.astype(np.float64)).astype(f32))
        band = fma(np.full(n, kd, f32), den32, fma(es2, np.abs(s32), c))
        return r <= -band, r >= band, tmid

    n = 2_000_000
    E = rng.normal(0, 1, (n, 9)); E /= np.linalg.norm(E, axis=1, keepdims=True)
    scale = rng.choice([0.3, 1.0, 4.0], n)[:, None]
    P = rng.uniform(-1, 1, (n, 4)) * scale                       # x1 y1 x2 y2 (double, as k_pose_prep leaves them)
    s32, den32, s, den, es, ed = replay(E, P)
    N1 = np.sqrt(P[:, 0] ** 2 + P[:, 1] ** 2 + 1.0); N2 = np.sqrt(P[:, 2] ** 2 + P[:, 3] ** 2 + 1.0)
    rs = np.abs(s32.astype(np.float64) - s) / (u * N1 * N2); rd = np.abs(den32.astype(np.float64) - den) / (u * (N1 * N1 + N2 * N2))
    assert rs.max() < 7 and rd.max() < 12, (rs.max(), rd.max())   # the derived bounds (the kernel's radii are 8 and 16)
    assert rs.max() > 1 and rd.max() > 1, (rs.max(), rd.max())    # ... and they are not vacuous
    for thr in (1.0 / 458.654, 0.25 / 458.654, 3.0 / 150.0):
        sure_in, sure_out, tmid = decide(s32, den32, es, ed, thr)
        truth = (s * s) <= tmid * den
        assert not (sure_in & ~truth).any() and not (sure_out & truth).any(), thr
        assert (sure_in | sure_out).mean() > 0.97                  # and single precision does decide nearly everything
        # pairs AT the threshold: y2 moved onto a root of s(y2)^2 = tmid den(y2) (a quadratic in y2), then off it by a relative
        # 10^-8 ... 10^-2 either way: most of these are undecided, none may be decided wrongly
        m = 300_000
        Em, Pm = E[:m], P[:m].copy()
        X1, Y1, X2 = Pm[:, 0], Pm[:, 1], Pm[:, 2]
        Ex0 = Em[:, 0] * X1 + Em[:, 1] * Y1 + Em[:, 2]; Ex1 = Em[:, 3] * X1 + Em[:, 4] * Y1 + Em[:, 5]; Ex2 = Em[:, 6] * X1 + Em[:, 7] * Y1 + Em[:, 8]
        a0, b0 = X2 * Ex0 + Ex2, Ex1
        c0, d0, c1, d1 = Em[:, 0] * X2 + Em[:, 6], Em[:, 3], Em[:, 1] * X2 + Em[:, 7], Em[:, 4]
        K = Ex0 ** 2 + Ex1 ** 2
        qa = b0 * b0 - tmid * (d0 * d0 + d1 * d1); qb = 2 * (a0 * b0 - tmid * (c0 * d0 + c1 * d1)); qc = a0 * a0 - tmid * (K + c0 * c0 + c1 * c1)
        disc = qb * qb - 4 * qa * qc
        ok = (disc > 0) & (np.abs(qa) > 1e-12)
        root = np.where(ok, (-qb + np.sqrt(np.where(ok, disc, 0))) / np.where(ok, 2 * qa, 1), 0.0)
        ok &= np.abs(root) < 4
        off = 10.0 ** rng.uniform(-8, -2, m) * rng.choice([-1.0, 1.0], m)
        Pm[:, 3] = root * (1 + off)
        Em, Pm = Em[ok], Pm[ok]
        assert len(Em) > 100_000
        t32, tden32, ts, tden, tes, ted = replay(Em, Pm)
        tin, tout, _ = decide(t32, tden32, tes, ted, thr)
        ttruth = (ts * ts) <= tmid * tden
        assert not (tin & ~ttruth).any() and not (tout & ttruth).any(), thr
        und = ~(tin | tout)
        assert 0.2 < und.mean() < 0.98, und.mean()                # both kinds occur: the check is not vacuous


def _np_half(src):
    """cv::resize(src, dst, Size(), 0.5, 0.5) as OpenCV 3.2 documents / implements it for an exact factor of 2 (resizeAreaFast_Invoker):
    dsize = cvRound(size * 0.5) (round half to even); a destination pixel is the mean of the source pixels of its 2 x 2 block that exist:
    (a + b + c + d + 2) >> 2 for a complete block, saturate_cast<uchar>((float)sum / count) -- round half to even -- otherwise."""
    h, w = src.shape
    rhe = lambda n: (n >> 1) + ((n & 1) & ((n >> 1) & 1))              # cvRound(n / 2): ties to even
    dh, dw = rhe(h), rhe(w)
    out = np.zeros((dh, dw), np.uint8)
    s = src.astype(np.int64)
    for y in range(dh):
        for x in range(dw):
            blk = s[2 * y:min(2 * y + 2, h), 2 * x:min(2 * x + 2, w)]
            if blk.size == 4:
                out[y, x] = (int(blk.sum()) + 2) >> 2
            else:
                out[y, x] = int(np.rint(np.float32(blk.sum()) / np.float32(blk.size)))      # numpy rint = ties to even, like cvRound
    return out


def test_half_pyramid_rule_in_numpy(orc):
    """the oracle's Camera::Update against the rule written out in numpy (sizes that do and do not halve exactly, incl. the image of the
    committed half_150x110 fixture): independent of oracle/orb.cpp, still not OpenCV itself (parity unpinned)"""
    import os
    rng = np.random.default_rng(11)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "half_150x110.npz"))
    imgs = [g["img"], rng.integers(0, 256, (47, 61), dtype=np.uint8), rng.integers(0, 256, (19, 18), dtype=np.uint8),
            rng.integers(0, 256, (64, 96), dtype=np.uint8), (rng.integers(0, 2, (33, 35)) * 255).astype(np.uint8)]
    for img in imgs:
        lv = orc.half_pyramid(img)
        ref = img
        for l in range(1, 5):
            ref = _np_half(ref)
            assert lv[l].shape == ref.shape, (img.shape, l, lv[l].shape, ref.shape)
            assert np.array_equal(lv[l], ref), (img.shape, l)
