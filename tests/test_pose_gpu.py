"""GPU parity: essential RANSAC / recoverPose / F2FRansac vs the CPU oracle.
Stated tolerance (floating point, FP64 on both sides, same algorithm, different libm):
  E (unit Frobenius norm, sign-normalised): max abs diff <= 1e-9
  R, t: max abs diff <= 1e-9 ; inlier mask, inlier count, iterations run: identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-9


def two_view(n, seed, outliers=0.0, noise=0.0, depth=(4.0, 12.0)):
    rng = np.random.default_rng(seed)
    K = np.array([[458.654, 0, 367.215], [0, 458.654, 248.375], [0, 0, 1]])
    ang = rng.normal(0, 0.05, 3)
    th = np.linalg.norm(ang)
    k = ang / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    t = rng.normal(0, 1, 3)
    t /= np.linalg.norm(t)
    X = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(*depth, n)], 1)
    x1 = (K @ X.T).T
    x1 = x1[:, :2] / x1[:, 2:]
    X2 = (R @ X.T).T + t
    x2 = (K @ X2.T).T
    x2 = x2[:, :2] / x2[:, 2:]
    x1 += rng.normal(0, noise, x1.shape)
    x2 += rng.normal(0, noise, x2.shape)
    nout = int(outliers * n)
    x2[:nout] = rng.uniform(0, 480, (nout, 2))
    return x1.astype(np.float32), x2.astype(np.float32), R, t


def _cmpE(E, oE):
    s = 1.0 if float((E * oE).sum()) >= 0 else -1.0
    return np.abs(E - s * oE).max()


@pytest.mark.parametrize("n,outl,noise,seed", [(49, 0.0, 0.0, 1), (49, 0.3, 0.3, 2), (200, 0.5, 0.5, 3), (30, 0.2, 0.2, 4), (6, 0, 0, 5), (5, 0, 0, 6), (4, 0, 0, 7)])
def test_essential_ransac_and_pose(vislam, orc, ctx, n, outl, noise, seed):
    p = vislam.default_params()
    p.fy = p.fx
    ctx.set_params(p)
    x1, x2, R, t = two_view(n, seed, outl, noise)
    E, mask, ninl, iters = ctx.essential_ransac(x1, x2)
    oE, omask, oninl, oiters = orc.essential_ransac(p, x1, x2)
    assert (ninl, iters) == (oninl, oiters)
    assert (mask == omask).all()
    if oninl == 0:
        assert np.abs(E).max() == 0
        return
    assert _cmpE(E, oE) <= TOL
    Rg, tg, ng = ctx.recover_pose(oE, x1, x2)
    Ro, to, no = orc.recover_pose(p, oE, x1, x2)
    assert ng == no
    assert np.abs(Rg - Ro).max() <= TOL and np.abs(tg - to).max() <= TOL
    if outl == 0.0 and n >= 30:       # sanity against ground truth
        assert np.abs(Ro - R).max() < 1e-3 and np.abs(to - t).max() < 1e-3


@pytest.mark.parametrize("seed,outl,noise,fx,thr,iters,adaptive", [(249849968, 0.0, 0.5, 150.0, 1.0, 300, 1), (653569463, 0.3, 0.5, 150.0, 0.25, 17, 1),
                                                                    (249849968, 0.0, 0.5, 150.0, 1.0, 300, 0)])
def test_five_correspondences_without_a_model(vislam, orc, ctx, seed, outl, noise, fx, thr, iters, adaptive):
    """count == modelPoints: a single solver run on all five points; when it finds no model the answer is "no E, no inliers, no
    iteration" (RANSACPointSetRegistrator::run returns false), not an all-ones mask around a stale model.  Found by
    tools/stress_pose.py (2 of 16954 random problems)."""
    p = vislam.default_params()
    p.fx = p.fy = fx
    p.ransac_threshold, p.ransac_prob, p.ransac_max_iters, p.ransac_adaptive = thr, 0.9, iters, adaptive
    ctx.set_params(p)
    x1, x2, R, t = two_view(5, seed, outl, noise)
    oE, omask, oninl, oiters = orc.essential_ransac(p, x1, x2)
    assert (oninl, oiters) == (0, 0)                              # the cases were picked for that
    E, mask, ninl, iters_run = ctx.essential_ransac(x1, x2)
    assert (ninl, iters_run) == (0, 0) and not mask.any() and np.abs(E).max() == 0


def test_ransac_fixed_iterations(vislam, orc, ctx):
    p = vislam.default_params()
    p.fy = p.fx
    p.ransac_adaptive = 0
    p.ransac_max_iters = 300
    ctx.set_params(p)
    x1, x2, R, t = two_view(400, 11, 0.4, 0.4)
    E, mask, ninl, iters = ctx.essential_ransac(x1, x2)
    oE, omask, oninl, oiters = orc.essential_ransac(p, x1, x2)
    assert iters == oiters == 300 and ninl == oninl and (mask == omask).all()
    assert _cmpE(E, oE) <= TOL


def _f2f_inputs(vislam, m, seed, outliers, noise):
    x1, x2, R, t = two_view(max(m, 5), seed, outliers, noise)
    KP = vislam.KEYPOINT_DTYPE
    a, b = np.zeros(m, KP), np.zeros(m, KP)
    a["x"], a["y"], b["x"], b["y"] = x1[:m, 0], x1[:m, 1], x2[:m, 0], x2[:m, 1]
    return a, b, R.T.astype(np.float32)


@pytest.mark.parametrize("m,seed,outl,noise,thr", [(40, 21, 0.1, 0.2, 370.0), (40, 5, 0.3, 0.5, 370.0), (120, 6, 0.5, 1.0, 370.0), (15, 7, 0.0, 0.0, 370.0),
                                                   (40, 8, 0.2, 0.3, 250.0), (40, 9, 0.2, 0.3, 600.0), (3, 10, 0.0, 0.1, 370.0), (2, 11, 0.0, 0.0, 370.0)])
def test_f2f_ransac(vislam, orc, ctx, m, seed, outl, noise, thr):
    """VISystem::F2FRansac (src/VISystem.cpp:612-769): the inlier test -1000 / log10(|d . n_i|) < threshold compares a
    transcendental (device libm vs host libm) against a constant, and the winner is the strictly larger COUNT: seeds, noise
    levels, thresholds and M down to 2 (M = 1 divides by zero in the reference; specified as the zero vector)"""
    p = vislam.default_params()
    p.f2f_threshold = thr
    ctx.set_params(p)
    a, b, rot = _f2f_inputs(vislam, m, seed, outl, noise)
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, max(m - 1, 1), (1000, 2)).astype(np.int32)       # rand() % (n-1), src/VISystem.cpp:712-713
    got, cg = ctx.f2f_ransac(a, b, rot, idx, 0.37)
    ref, co = orc.f2f_ransac(p, a, b, rot, idx, 0.37)
    assert cg == co, (cg, co)
    assert np.abs(got - ref).max() <= 1e-6


def test_f2f_ransac_counts_near_a_tie(vislam, orc, ctx):
    """thresholds swept through the value at which the best hypothesis's count changes: whatever side of a boundary a
    correspondence falls on, both implementations must put it on the same side (the count compare is exact)"""
    p = vislam.default_params()
    a, b, rot = _f2f_inputs(vislam, 60, 33, 0.25, 0.4)
    idx = np.random.default_rng(33).integers(0, 59, (1000, 2)).astype(np.int32)
    seen = set()
    for thr in np.linspace(150.0, 900.0, 26):
        p.f2f_threshold = float(thr)
        ctx.set_params(p)
        got, cg = ctx.f2f_ransac(a, b, rot, idx, 1.0)
        ref, co = orc.f2f_ransac(p, a, b, rot, idx, 1.0)
        assert cg == co, (thr, cg, co)
        assert np.abs(got - ref).max() <= 1e-6
        seen.add(cg)
    assert len(seen) >= 5                                          # the sweep really crossed count boundaries


def test_f2f_ransac_degenerate(vislam, ctx):
    ctx.set_params(vislam.default_params())
    KP = vislam.KEYPOINT_DTYPE
    z, c0 = ctx.f2f_ransac(np.zeros(1, KP), np.zeros(1, KP), np.eye(3, dtype=np.float32), np.zeros((0, 2), np.int32), 1.0)
    assert (z == 0).all() and c0 == 0


def test_batch_pipeline_pose(vislam, orc, canvas):
    """end-to-end batched path: detect -> match -> pose vs the oracle's per-frame pipeline"""
    import torch
    p = vislam.default_params()
    p.fy = p.fx
    c = vislam.Context(0, p)
    n = 6
    frames = np.stack([vislam.synth_frame(canvas, t, 752, 480) for t in range(n)])
    dev = torch.from_numpy(frames).cuda()
    c.batch_plan(752, 480, 752, n)
    c.batch_run(dev.data_ptr(), n)
    c.batch_sync()
    assert c.batch_status() == 0
    prev = None
    for t in range(n):
        ok, od, r = orc.pipeline_frame(p, frames[t], prev)
        prev = (ok, od)
        g, nsym = c.batch_matches(t)
        pose = c.batch_pose(t)
        assert nsym == r.n_sym and len(g) == r.n_good
        assert pose["n_inliers"] == r.n_inliers and pose["iters_run"] == r.iters_run, t
        if r.n_inliers:
            oE = np.array(r.E).reshape(3, 3)
            assert _cmpE(pose["E"], oE) <= TOL
            assert pose["n_pose_good"] == r.n_pose_good
            assert np.abs(pose["R"] - np.array(r.R).reshape(3, 3)).max() <= 1e-7
            assert np.abs(pose["t"] - np.array(r.t)).max() <= 1e-7
    c.close()


def test_recycled_device_memory_does_not_leak_into_results(vislam, orc, canvas):
    """regression: lanes of a RANSAC wave whose pair has no work (first frame of a stream: M = 0) once read
    never-written sample slots; with recycled device memory that became an out-of-bounds index.  Poison the
    allocator with a context that is destroyed, then run a fresh stream and check parity."""
    import torch
    p = vislam.default_params()
    p.fy = p.fx
    rng = np.random.default_rng(0)
    a = vislam.Context(0, p)
    noise = torch.from_numpy(rng.integers(0, 256, (8, 480, 752), dtype=np.uint8)).cuda()
    a.batch_plan(752, 480, 752, 8)
    a.batch_run(noise.data_ptr(), 8)
    a.batch_sync()
    a.close()
    c = vislam.Context(0, p)
    frames = np.stack([vislam.synth_frame(canvas, t, 752, 480) for t in range(5)])
    dev = torch.from_numpy(frames).cuda()
    c.batch_plan(752, 480, 752, 8)
    c.batch_run(dev.data_ptr(), 5)
    c.batch_sync()
    assert c.batch_status() == 0
    prev = None
    for t in range(5):
        ok, od, r = orc.pipeline_frame(p, frames[t], prev)
        prev = (ok, od)
        pose = c.batch_pose(t)
        assert pose["n_inliers"] == r.n_inliers and pose["iters_run"] == r.iters_run
    c.close()


@pytest.mark.parametrize("m,thr,fx,noise,adaptive", [(300, 1.0, 458.654, 1.0, 1), (1000, 0.25, 458.654, 0.3, 1), (3000, 1.0, 150.0, 1.0, 1),
                                                     (3000, 3.0, 458.654, 3.0, 0), (5000, 1.0, 458.654, 0.7, 1), (2049, 0.5, 90.0, 0.5, 0)])
def test_many_correspondences_single_precision_scoring(vislam, orc, ctx, m, thr, fx, noise, adaptive):
    """k_hyp_score's many-correspondences form decides the Sampson test in single precision where its error radii allow and in the oracle's
    double sequence elsewhere (pose.hip).  Pixel noise of the size of the threshold puts thousands of residuals next to it; a short
    focal length makes the normalised coordinates (and the radii) large; 2049 points exercise the row dealing (9 rows as 3 + 3 + 3).
    Masks, counts and iteration numbers must be the oracle's, E within the stated tolerance."""
    p = vislam.default_params()
    p.fx = p.fy = fx
    p.ransac_threshold, p.ransac_adaptive, p.ransac_max_iters = thr, adaptive, 400
    ctx.set_params(p)
    rng = np.random.default_rng(m + int(10 * thr))
    x1, x2, R, t = two_view(m, 100 + m, 0.35, 0.0)
    x2 = (x2 + rng.normal(0, noise, x2.shape)).astype(np.float32)
    E, mask, ninl, iters = ctx.essential_ransac(x1, x2)
    oE, omask, oninl, oiters = orc.essential_ransac(p, x1, x2)
    assert (ninl, iters) == (oninl, oiters)
    assert (mask == omask).all()
    assert 100 < oninl < 0.95 * m                            # the threshold really cuts through the residual distribution (the short focal
                                                             # lengths do not match the K the points were projected with: few inliers, large coordinates)
    assert _cmpE(E, oE) <= TOL
