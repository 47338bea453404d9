"""GPU: the BASELINE.json configurations beside the headline one.
 config 1: the reference CPU path's own ORB::create(200) (src/Camera.cpp:127) at 752x480, on the HIP path
 config 3: 1920x1080, 4 levels, 4000 kps + essential RANSAC with a fixed 2000 iterations
 config 5: 3840x2160, 8000 kps, 8000x8000 all-pairs
Parity against the oracle where it finishes in seconds, size-independent properties otherwise."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config1_cpu_default_orb200_752x480(vislam, orc, canvas):
    """BASELINE config 1 (stand-in for EuRoC MH_01: the dataset is not in the image): the CPU main's detector
    setting nfeatures = 200, first frames of S-752, single-frame API and the batched stream path, both
    bit-exact against the oracle's per-frame pipeline (keypoints, descriptors, matches) and pose within 1e-7."""
    import torch
    p = vislam.default_params()
    p.nfeatures = 200
    p.fy = p.fx
    c = vislam.Context(0, p)
    ws, hs, sc, q = c.level_geometry(752, 480)
    assert list(q) == [43, 36, 30, 25, 21, 17, 15, 13]               # SURVEY 8(a) a4 quotas @200
    n = 8
    frames = np.stack([vislam.synth_frame(canvas, t, 752, 480) for t in range(n)])
    dev = torch.from_numpy(frames).cuda()
    c.batch_plan(752, 480, 752, n)
    c.batch_run(dev.data_ptr(), n)
    c.batch_sync()
    assert c.batch_status() == 0
    prev = None
    for t in range(n):
        ok, od, r = orc.pipeline_frame(p, frames[t], prev)
        assert 180 <= len(ok) <= 260
        k, d = c.batch_keypoints(t)
        assert k.tobytes() == ok.tobytes() and (d == od).all(), t
        if t in (0, 5):                                                # the single-frame entry point on the same frames
            k1, d1 = c.orb_detect_compute(frames[t], slot=t)
            assert k1.tobytes() == ok.tobytes() and (d1 == od).all()
        g, nsym = c.batch_matches(t)
        pose = c.batch_pose(t)
        assert nsym == r.n_sym and len(g) == r.n_good
        if prev is not None:
            o12, o21 = orc.knn2_hamming(prev[1], od)
            og, osym = orc.good_matches(p, prev[0], ok, o12, o21)
            assert g.tobytes() == og.tobytes()
            g12, g21 = c.batch_knn(t)
            assert g12.tobytes() == o12.tobytes() and g21.tobytes() == o21.tobytes()
        assert pose["n_inliers"] == r.n_inliers and pose["iters_run"] == r.iters_run, t
        if r.n_inliers:
            oE = np.array(r.E).reshape(3, 3)
            s = 1.0 if float((pose["E"] * oE).sum()) >= 0 else -1.0
            assert np.abs(pose["E"] - s * oE).max() <= 1e-9
            assert pose["n_pose_good"] == r.n_pose_good
            assert np.abs(pose["R"] - np.array(r.R).reshape(3, 3)).max() <= 1e-7
        prev = (ok, od)
    c.close()


@pytest.fixture(scope="module")
def big_canvas(vislam):
    return vislam.synth_canvas(4096, 0xE0C00003)


def test_config3_1080p_4levels_4000kps(vislam, orc, big_canvas):
    p = vislam.default_params()
    p.nfeatures, p.nlevels, p.w_size, p.h_size = 4000, 4, 1920, 1080
    p.ransac_adaptive, p.ransac_max_iters = 0, 2000
    p.fy = p.fx
    c = vislam.Context(0, p)
    a = vislam.synth_frame(big_canvas, 0, 1920, 1080, 0xE0C00003)
    b = vislam.synth_frame(big_canvas, 1, 1920, 1080, 0xE0C00003)
    k0, d0 = c.orb_detect_compute(a, slot=0, cap=8192)
    k1, d1 = c.orb_detect_compute(b, slot=1, cap=8192)
    ok0, od0 = orc.orb_detect_compute(p, a, cap=8192)
    assert len(k0) >= 3900 and k0.tobytes() == ok0.tobytes() and (d0 == od0).all()
    g12, g21 = c.bf_knn2_hamming(0, 1, len(k0), len(k1))
    o12, o21 = orc.knn2_hamming(d0, d1)
    assert g12.tobytes() == o12.tobytes() and g21.tobytes() == o21.tobytes()
    good, sym = c.good_matches(0, 1)
    og, osym = orc.good_matches(p, k0, k1, o12, o21)
    assert good.tobytes() == og.tobytes() and sym.tobytes() == osym.tobytes() and len(sym) > 1500
    # RANSAC on the un-gridded symmetric matches (M ~ thousands), 2000 fixed iterations
    p1 = np.stack([k0["x"][sym["queryIdx"]], k0["y"][sym["queryIdx"]]], 1)
    p2 = np.stack([k1["x"][sym["trainIdx"]], k1["y"][sym["trainIdx"]]], 1)
    E, mask, ninl, iters = c.essential_ransac(p1, p2)
    oE, omask, oninl, oiters = orc.essential_ransac(p, p1, p2)
    assert iters == oiters == 2000 and ninl == oninl and (mask == omask).all()
    s = 1.0 if float((E * oE).sum()) >= 0 else -1.0
    assert np.abs(E - s * oE).max() <= 1e-9
    c.close()


def test_config5_2160p_8000kps_properties(vislam, orc, big_canvas):
    p = vislam.default_params()
    p.nfeatures, p.nlevels, p.w_size, p.h_size = 8000, 8, 3840, 2160
    c = vislam.Context(0, p)
    a = vislam.synth_frame(big_canvas, 0, 3840, 2160, 0xE0C00003)
    k0, d0 = c.orb_detect_compute(a, slot=0, cap=16384)
    k1, d1 = c.orb_detect_compute(a, slot=1, cap=16384)          # same image twice: idempotence
    assert len(k0) >= 7900 and k0.tobytes() == k1.tobytes() and (d0 == d1).all()
    ws, hs, sc, q = c.level_geometry(3840, 2160)
    # canonical order: octave ascending, response descending inside an octave; per-level counts >= quota unless starved
    assert (np.diff(k0["octave"]) >= 0).all()
    for l in range(8):
        r = k0["response"][k0["octave"] == l]
        assert (np.diff(r) <= 0).all() and len(r) >= q[l]
    # keypoints respect the 31 px border in their own level
    x_l = k0["x"] / sc[k0["octave"]]
    y_l = k0["y"] / sc[k0["octave"]]
    assert (x_l >= 30.5).all() and (x_l < ws[k0["octave"]] - 30.5).all() and (y_l >= 30.5).all() and (y_l < hs[k0["octave"]] - 30.5).all()
    # 8000 x 8000 all-pairs on identical sets: every row's best match is itself at distance 0
    g12, g21 = c.bf_knn2_hamming(0, 1, len(k0), len(k1))
    assert (g12["distance"][:, 0] == 0).all() and (g12["distance"][:, 0] <= g12["distance"][:, 1]).all()
    dup = g12["trainIdx"][:, 0] != np.arange(len(k0))
    assert (g12["distance"][dup, 1] == 0).all()                 # only exact duplicates may rank a lower index first
    assert g12.tobytes() == g21.tobytes()
    # oracle parity on the full frame is still affordable once
    ok0, od0 = orc.orb_detect_compute(p, a, cap=16384)
    assert k0.tobytes() == ok0.tobytes() and (d0 == od0).all()
    good, sym = c.good_matches(0, 1)
    assert len(sym) > 7000 and len(good) == 49
    c.close()


def test_batched_pose_on_symmetric_matches(vislam, orc, canvas):
    """BASELINE config 3 runs RANSAC on the un-gridded symmetric matches (M in the hundreds or thousands, where the
    reference pipeline feeds at most root^2 = 49): vis_params.pose_input = VIS_POSE_SYM in the batched path, fixed
    iteration count, checked per pair against the oracle's findEssentialMat / recoverPose on the same matches."""
    import torch
    p = vislam.default_params()
    p.fy = p.fx
    p.pose_input = 1                                                  # VIS_POSE_SYM
    p.ransac_adaptive, p.ransac_max_iters = 0, 150
    c = vislam.Context(0, p)
    n = 4
    frames = np.stack([vislam.synth_frame(canvas, t, 752, 480) for t in range(n)])
    dev = torch.from_numpy(frames).cuda()
    c.batch_plan(752, 480, 752, n)
    c.batch_run(dev.data_ptr(), n)
    c.batch_sync()
    assert c.batch_status() == 0
    kd = [c.batch_keypoints(t) for t in range(n)]
    for t in range(1, n):
        o12, o21 = orc.knn2_hamming(kd[t - 1][1], kd[t][1])
        og, osym = orc.good_matches(p, kd[t - 1][0], kd[t][0], o12, o21)
        g, nsym = c.batch_matches(t)
        assert nsym == len(osym) > 300 and g.tobytes() == og.tobytes()
        p1 = np.stack([kd[t - 1][0]["x"][osym["queryIdx"]], kd[t - 1][0]["y"][osym["queryIdx"]]], 1)
        p2 = np.stack([kd[t][0]["x"][osym["trainIdx"]], kd[t][0]["y"][osym["trainIdx"]]], 1)
        oE, omask, oninl, oiters = orc.essential_ransac(p, p1, p2)
        pose = c.batch_pose(t)
        assert pose["iters_run"] == oiters == 150 and pose["n_inliers"] == oninl, t
        s = 1.0 if float((pose["E"] * oE).sum()) >= 0 else -1.0
        assert np.abs(pose["E"] - s * oE).max() <= 1e-9
        oR, ot, ong = orc.recover_pose(p, oE, p1, p2)
        assert pose["n_pose_good"] == ong and np.abs(pose["R"] - oR).max() <= 1e-7
    c.close()


def test_config3_batched_path_that_the_bench_times(vislam, orc, big_canvas):
    """bench.py's config-3 leg: a BATCH of 1920x1080 frames, 4 levels, 4000 keypoints, RANSAC with 2000 fixed iterations on the
    un-gridded symmetric matches (pose_input = SYM), through vis_batch_run(STAGE_FRAME) -- the launch sequence the leg times, not the
    single-frame entry points -- against the oracle per frame / pair: keypoints, descriptors, both 2-NN tables, symmetric and good
    matches, inlier masks, inlier counts and iteration numbers exact; E <= 1e-9, R <= 1e-7 (the batch tolerances of DESIGN.md section 2)."""
    import torch
    p = vislam.default_params()
    p.nfeatures, p.nlevels, p.w_size, p.h_size = 4000, 4, 1920, 1080
    p.ransac_adaptive, p.ransac_max_iters = 0, 2000
    p.pose_input = 1                                                  # VIS_POSE_SYM
    p.fy = p.fx
    c = vislam.Context(0, p)
    n = 8
    frames = np.stack([vislam.synth_frame(big_canvas, t, 1920, 1080, 0xE0C00003) for t in range(n)])
    dev = torch.from_numpy(frames).cuda()
    c.batch_plan(1920, 1080, 1920, n)
    c.batch_run(dev.data_ptr(), n, vislam.STAGE_FRAME)
    c.batch_sync()
    assert c.batch_status() == 0
    # Camera::Update of the same launch (the half pyramid of every frame)
    dptr, fe = c.batch_half_pyramid()
    class _Dev:                                                        # the plan's buffer as a torch view
        __cuda_array_interface__ = {"data": (dptr, False), "shape": (n * fe,), "typestr": "|u1", "version": 2}
    half = torch.as_tensor(_Dev(), device="cuda").cpu().numpy().reshape(n, fe)
    for t in (0, n - 1):
        ref = orc.half_pyramid(frames[t])
        off = 1920 * 1080
        for l in range(1, 5):
            assert np.array_equal(half[t, off:off + ref[l].size].reshape(ref[l].shape), ref[l]), (t, l)
            off += ref[l].size
    prev = None
    for t in range(n):
        ok, od = orc.orb_detect_compute(p, frames[t], cap=8192)
        k, d = c.batch_keypoints(t, cap=8192)
        assert len(k) >= 3900 and k.tobytes() == ok.tobytes() and (d == od).all(), t
        if prev is not None:
            o12, o21 = orc.knn2_hamming(prev[1], od)
            g12, g21 = c.batch_knn(t, cap=8192)
            assert g12.tobytes() == o12.tobytes() and g21.tobytes() == o21.tobytes(), t
            og, osym = orc.good_matches(p, prev[0], ok, o12, o21)
            g, nsym = c.batch_matches(t)
            assert nsym == len(osym) > 1500 and g.tobytes() == og.tobytes(), t
            p1 = np.stack([prev[0]["x"][osym["queryIdx"]], prev[0]["y"][osym["queryIdx"]]], 1)
            p2 = np.stack([ok["x"][osym["trainIdx"]], ok["y"][osym["trainIdx"]]], 1)
            oE, omask, oninl, oiters = orc.essential_ransac(p, p1, p2)
            pose = c.batch_pose(t)
            mask = c.batch_inlier_mask(t)
            assert pose["iters_run"] == oiters == 2000 and pose["n_inliers"] == oninl, t
            assert len(mask) == len(omask) and (mask == omask).all(), t
            s = 1.0 if float((pose["E"] * oE).sum()) >= 0 else -1.0
            assert np.abs(pose["E"] - s * oE).max() <= 1e-9, t
            oR, ot, ong = orc.recover_pose(p, oE, p1, p2)
            assert pose["n_pose_good"] == ong and np.abs(pose["R"] - oR).max() <= 1e-7, t
        prev = (ok, od)
    c.close()


def test_config5_8000x8000_both_directions_and_exact_filters(vislam, orc, big_canvas):
    """BASELINE config 5 at full size on TWO DIFFERENT frames: sampled rows of BOTH 2-NN tables (g12 and the transposed pass g21)
    against a numpy popcount over all 8000 train descriptors, and Matcher's filter chain (ratio, symmetry, y sort, grid cells: O(N) on
    the CPU) on the GPU's own tables -- symmetric and good matches exact at n = 8000 (round 4 checked counts only)."""
    p = vislam.default_params()
    p.nfeatures, p.nlevels, p.w_size, p.h_size = 8000, 8, 3840, 2160
    c = vislam.Context(0, p)
    a = vislam.synth_frame(big_canvas, 0, 3840, 2160, 0xE0C00003)
    b = vislam.synth_frame(big_canvas, 1, 3840, 2160, 0xE0C00003)
    k0, d0 = c.orb_detect_compute(a, slot=0, cap=16384)
    k1, d1 = c.orb_detect_compute(b, slot=1, cap=16384)
    assert len(k0) >= 7900 and len(k1) >= 7900
    g12, g21 = c.bf_knn2_hamming(0, 1, len(k0), len(k1))
    assert len(g12) == len(k0) and len(g21) == len(k1)
    bits0, bits1 = np.unpackbits(d0, axis=1), np.unpackbits(d1, axis=1)
    rng = np.random.default_rng(5)

    def check(tab, qbits, tbits, nq):
        rows = np.unique(np.concatenate([[0, 1, nq - 1, nq - 2], rng.integers(0, nq, 96)]))
        dist = (qbits[rows][:, None, :] != tbits[None, :, :]).sum(2)             # rows x n_train Hamming distances
        order = np.argsort(dist, axis=1, kind="stable")                           # ties: lower train index first (BFMatcher's scan order)
        for i, r in enumerate(rows):
            assert tab["trainIdx"][r, 0] == order[i, 0] and tab["trainIdx"][r, 1] == order[i, 1], (r, tab[r], order[i, :2])
            assert tab["distance"][r, 0] == dist[i, order[i, 0]] and tab["distance"][r, 1] == dist[i, order[i, 1]]
            assert tab["queryIdx"][r, 0] == r and tab["queryIdx"][r, 1] == r
    check(g12, bits0, bits1, len(k0))
    check(g21, bits1, bits0, len(k1))
    good, sym = c.good_matches(0, 1)
    og, osym = orc.good_matches(p, k0, k1, g12, g21)                              # the oracle's filters on the GPU's tables
    assert sym.tobytes() == osym.tobytes() and good.tobytes() == og.tobytes()
    assert len(sym) > 2000 and 0 < len(good) <= 49
    c.close()


def test_config5_batched_path_that_the_bench_times(vislam, orc, big_canvas):
    """bench.py's config-5 leg: a BATCH of 3840x2160 frames, 8 levels, 8000 keypoints through vis_batch_run(STAGE_FRAME) -- the launch
    sequence the leg times (multi-frame indexing at that size: ~26 M pyramid pixels per frame, k_select_1024, k_filter<1024>, the carried
    frame), not the single-frame entry points.  Keypoints + descriptors of the first, the second-to-last and the last frame exact against
    the oracle; Camera::Update's half pyramid of the last frame exact; both 2-NN tables of the last pair on >= 100 sampled rows against a
    numpy popcount over all train descriptors; symmetric / good matches exact (the oracle's filter chain on the GPU's own tables); the
    pose record against the oracle's essential RANSAC + recoverPose on those good matches (E <= 1e-9, R <= 1e-7: DESIGN.md section 2).
    A second launch continues the stream: its frame 0 is matched against the carried last frame of the first launch."""
    import torch
    p = vislam.default_params()
    p.nfeatures, p.nlevels, p.w_size, p.h_size = 8000, 8, 3840, 2160
    p.fy = p.fx
    c = vislam.Context(0, p)
    n = 5
    frames = np.stack([vislam.synth_frame(big_canvas, t, 3840, 2160, 0xE0C00003) for t in range(n + 1)])
    dev = torch.from_numpy(frames).cuda()
    c.batch_plan(3840, 2160, 3840, n)
    c.batch_run(dev.data_ptr(), n, vislam.STAGE_FRAME)
    c.batch_sync()
    assert c.batch_status() == 0
    ref = {}
    for t in (0, n - 2, n - 1):
        ref[t] = orc.orb_detect_compute(p, frames[t], cap=16384)
        k, d = c.batch_keypoints(t, cap=16384)
        assert len(k) >= 7900 and k.tobytes() == ref[t][0].tobytes() and (d == ref[t][1]).all(), t
    dptr, fe = c.batch_half_pyramid()
    class _Dev:
        __cuda_array_interface__ = {"data": (dptr, False), "shape": (n * fe,), "typestr": "|u1", "version": 2}
    half = torch.as_tensor(_Dev(), device="cuda")[(n - 1) * fe:n * fe].cpu().numpy()
    off = 3840 * 2160
    for l, lv in enumerate(orc.half_pyramid(frames[n - 1])):
        if l:
            assert np.array_equal(half[off:off + lv.size].reshape(lv.shape), lv), l
            off += lv.size

    def check_pair(t, kq, dq, kt, dt):
        g12, g21 = c.batch_knn(t, cap=16384)
        assert len(g12) == len(kq) and len(g21) == len(kt)
        bq, bt = np.unpackbits(dq, axis=1), np.unpackbits(dt, axis=1)
        rng = np.random.default_rng(50 + t)
        for tab, qb, tb, nq in ((g12, bq, bt, len(kq)), (g21, bt, bq, len(kt))):
            rows = np.unique(np.concatenate([[0, 1, nq - 1, nq - 2], rng.integers(0, nq, 100)]))
            dist = (qb[rows][:, None, :] != tb[None, :, :]).sum(2)
            order = np.argsort(dist, axis=1, kind="stable")                       # ties: lower train index first (BFMatcher's scan order)
            for i, r in enumerate(rows):
                assert tab["trainIdx"][r, 0] == order[i, 0] and tab["trainIdx"][r, 1] == order[i, 1], (t, r)
                assert tab["distance"][r, 0] == dist[i, order[i, 0]] and tab["distance"][r, 1] == dist[i, order[i, 1]], (t, r)
        og, osym = orc.good_matches(p, kq, kt, g12, g21)                          # the oracle's filters on the GPU's tables
        g, nsym = c.batch_matches(t)
        assert nsym == len(osym) > 2000 and g.tobytes() == og.tobytes() and 0 < len(g) <= 49, t
        p1 = np.stack([kq["x"][og["queryIdx"]], kq["y"][og["queryIdx"]]], 1)
        p2 = np.stack([kt["x"][og["trainIdx"]], kt["y"][og["trainIdx"]]], 1)
        oE, omask, oninl, oiters = orc.essential_ransac(p, p1, p2)
        pose = c.batch_pose(t)
        assert pose["iters_run"] == oiters and pose["n_inliers"] == oninl, t
        if oninl:
            s = 1.0 if float((pose["E"] * oE).sum()) >= 0 else -1.0
            assert np.abs(pose["E"] - s * oE).max() <= 1e-9, t
            oR, ot, ong = orc.recover_pose(p, oE, p1, p2)
            assert pose["n_pose_good"] == ong and np.abs(pose["R"] - oR).max() <= 1e-7, t
    check_pair(n - 1, ref[n - 2][0], ref[n - 2][1], ref[n - 1][0], ref[n - 1][1])
    # the stream goes on: one more frame in a second launch, matched against the carried frame n - 1
    c.batch_run(dev.data_ptr() + n * 3840 * 2160, 1, vislam.STAGE_FRAME)
    c.batch_sync()
    assert c.batch_status() == 0
    kn, dn = c.batch_keypoints(0, cap=16384)
    okn, odn = orc.orb_detect_compute(p, frames[n], cap=16384)
    assert kn.tobytes() == okn.tobytes() and (dn == odn).all()
    check_pair(0, ref[n - 1][0], ref[n - 1][1], okn, odn)
    c.close()
