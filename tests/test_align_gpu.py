"""GPU parity: VISystem::EstimatePoseFeatures (Gauss-Newton photometric alignment, src/VISystem.cpp:1113-1448) through the
C ABI vs the CPU oracle.  Both sides use the same operation order, the same deterministic sin/cos and the same
summation tree, so the comparison is BIT-EXACT (iteration counts, residual counts, errors and the 7 pose floats);
the tolerance that would apply against the real reference (OpenCV gemm / Sophus / libm) is unpinned."""
import ctypes as C
import os

import numpy as np
import pytest

import align_cases

pytestmark = pytest.mark.gpu


def _same(a, b):
    ta, tb = align_cases.result_tuple(a), align_cases.result_tuple(b)
    assert ta[0] == tb[0], ("iterations", ta[0], tb[0])
    assert ta[1] == tb[1], ("n_residuals", ta[1], tb[1])
    assert np.array_equal(np.array(ta[2], np.float32), np.array(tb[2], np.float32)), ("error", ta[2], tb[2])
    assert np.array_equal(np.array(ta[4], np.float32), np.array(tb[4], np.float32)), ("pose", ta[4], tb[4])
    assert np.array_equal(np.array(ta[5], np.float32), np.array(tb[5], np.float32))
    assert ta[3] == tb[3]


@pytest.mark.parametrize("dx,dy,div,n", [(0, 0, 1, 20), (2, 1, 1, 49), (4, -3, 1, 49), (4, 3, 8, 30), (2, 1, 16, 49), (1, 0, 24, 12),
                                         (8, 6, 4, 49), (3, -2, 96, 49), (2, 1, 400, 10)])
def test_single_pair_bit_exact(vislam, orc, ctx, canvas, dx, dy, div, n):
    c = align_cases.case(vislam, orc, canvas, dx=dx, dy=dy, n=n, grad_div=div)
    ap = vislam.default_align_params()
    got = ctx.estimate_pose_features(ap, 752, 480, c["gray1"], c["gray2"], c["gx"], c["gy"], c["cand"])
    ref = orc.estimate_pose_features(orc.default_align_params(), 752, 480, c["gray1"], c["gray2"], c["gx"], c["gy"], c["cand"])
    _same(got, ref)


@pytest.mark.parametrize("w,h", [(1080, 540), (600, 270), (270, 150)])
def test_single_pair_on_sizes_that_do_not_halve_exactly(vislam, orc, ctx, canvas, w, h):
    """levels one row / column larger than the reference's `size >> lvl` bookkeeping (270 -> 135 -> 68 against 67): coordinates are
    bounded by the bookkeeping sizes, strides and the clamp of the rounded index follow the level's own size (oracle/align.cpp)"""
    c = align_cases.case(vislam, orc, canvas, w=w, h=h, dx=3, dy=2, n=49, grad_div=8)
    lw, lh = vislam.half_pyramid_dims(w, h)
    assert [g.shape for g in c["gray1"]] == [(lh[l], lw[l]) for l in range(5)]
    assert any(lh[l] != (h >> l) or lw[l] != (w >> l) for l in range(5))
    ap = vislam.default_align_params(); oap = orc.default_align_params()
    for q in (ap, oap):
        q.fx, q.fy, q.cx, q.cy = 300.0, 300.0, w / 2.0, h / 2.0
    got = ctx.estimate_pose_features(ap, w, h, c["gray1"], c["gray2"], c["gx"], c["gy"], c["cand"])
    ref = orc.estimate_pose_features(oap, w, h, c["gray1"], c["gray2"], c["gx"], c["gy"], c["cand"])
    _same(got, ref)
    assert sum(ref.n_residuals) > 0


def test_options_levels_and_initial_pose(vislam, orc, ctx, canvas):
    c = align_cases.case(vislam, orc, canvas, dx=4, dy=3, n=40, grad_div=8)
    for first, last, iters, init6 in [(3, 0, 10, None), (2, 1, 4, [0.002, -0.001, 0, 0, 0, 0.001]), (4, 4, 3, [0, 0, 0, 1e-4, 2e-4, 0]),
                                      (0, 0, 25, [0.01, 0.01, 0, 0, 0, 0]), (3, 0, 1, None)]:
        ap = vislam.default_align_params()
        ap.first_level, ap.last_level, ap.max_iterations = first, last, iters
        init = None if init6 is None else orc.se3_exp(init6)
        got = ctx.estimate_pose_features(ap, 752, 480, c["gray1"], c["gray2"], c["gx"], c["gy"], c["cand"], init)
        oap = orc.default_align_params()
        oap.first_level, oap.last_level, oap.max_iterations = first, last, iters
        ref = orc.estimate_pose_features(oap, 752, 480, c["gray1"], c["gray2"], c["gx"], c["gy"], c["cand"], init)
        _same(got, ref)


def test_degenerate_inputs(vislam, orc, ctx, canvas):
    c = align_cases.case(vislam, orc, canvas, dx=1, dy=1, n=10)
    ap = vislam.default_align_params()
    init = orc.se3_exp([50.0, 0, 0, 0, 0, 0])                       # every point leaves the image: zero residuals
    _same(ctx.estimate_pose_features(ap, 752, 480, c["gray1"], c["gray2"], c["gx"], c["gy"], c["cand"], init),
          orc.estimate_pose_features(orc.default_align_params(), 752, 480, c["gray1"], c["gray2"], c["gx"], c["gy"], c["cand"], init))
    empty = [np.zeros((0, 4), np.float32)] * 5
    r = ctx.estimate_pose_features(ap, 752, 480, c["gray1"], c["gray2"], c["gx"], c["gy"], empty)
    assert list(r.n_residuals) == [0] * 5 and list(r.pose.as_array()) == [0, 0, 0, 1, 0, 0, 0]
    flat = [np.zeros_like(g) for g in c["gx"]]                      # zero gradients: singular normal matrix -> zero step
    _same(ctx.estimate_pose_features(ap, 752, 480, c["gray1"], c["gray2"], flat, flat, c["cand"]),
          orc.estimate_pose_features(orc.default_align_params(), 752, 480, c["gray1"], c["gray2"], flat, flat, c["cand"]))
    with pytest.raises(vislam.VisError):
        bad = vislam.default_align_params(); bad.first_level = 5
        ctx.estimate_pose_features(bad, 752, 480, c["gray1"], c["gray2"], c["gx"], c["gy"], c["cand"])


def test_golden_fixture(vislam, ctx):
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "align_320x240.npz"))
    ap = vislam.default_align_params()
    ap.fx, ap.fy, ap.cx, ap.cy = 200.0, 200.0, 160.0, 120.0
    lv = lambda k: [g[f"{k}{l}"] for l in range(5)]
    r = ctx.estimate_pose_features(ap, 320, 240, lv("gray1_"), lv("gray2_"), lv("gx_"), lv("gy_"), lv("cand_"))
    assert list(r.iterations) == list(g["iterations"]) and list(r.n_residuals) == list(g["n_residuals"])
    assert np.array_equal(np.array(r.error, np.float32), g["error"]) and np.array_equal(r.pose.as_array(), g["pose"])
    assert np.array_equal(np.array(r.matrix, np.float32), g["matrix"])


def test_batched_stream_alignment(vislam, orc, canvas):
    """the throughput path: detect -> match on n resident frames, half pyramid + gradients on the device, then ONE launch
    aligns every consecutive pair with the candidate pixels generated on the fly from the plan's good matches.
    Every pair must equal the oracle run on that pair's host-side data (the reference's per-frame sequence:
    addGPUKeyframe -> ObtainPatchesPointsPreviousFrame -> EstimatePoseFeatures)."""
    import torch
    W, H, n = 752, 480, 6
    p = vislam.default_params()
    c = vislam.Context(0, p)
    frames = np.stack([vislam.synth_frame(canvas, t, W, H) for t in range(n)])
    dev = torch.from_numpy(frames).cuda()
    c.batch_plan(W, H, W, n)
    c.batch_run(dev.data_ptr(), n, vislam.STAGE_DETECT | vislam.STAGE_MATCH)
    fe = vislam.gradient_frame_elems(W, H)
    gray = torch.zeros(n * fe, dtype=torch.uint8, device="cuda")
    gx = torch.zeros(n * fe, dtype=torch.int16, device="cuda"); gy = torch.zeros_like(gx)
    g = torch.zeros(n * fe, dtype=torch.uint8, device="cuda")
    c.gradient_batch(dev.data_ptr(), W, H, W, n, gray.data_ptr(), gx.data_ptr(), gy.data_ptr(), g.data_ptr())
    out = torch.zeros(n * C.sizeof(vislam.AlignResult), dtype=torch.uint8, device="cuda")
    ap = vislam.default_align_params()
    c.batch_align(ap, dev.data_ptr(), n, gray.data_ptr(), gx.data_ptr(), gy.data_ptr(), 0, out.data_ptr())
    c.batch_sync()
    torch.cuda.synchronize()
    raw = out.cpu().numpy().tobytes()
    res = [vislam.AlignResult.from_buffer_copy(raw, i * C.sizeof(vislam.AlignResult)) for i in range(n)]
    assert list(res[0].n_residuals) == [0] * 5                       # frame 0 has no predecessor in the batch
    kd = [c.batch_keypoints(t) for t in range(n)]
    npairs_checked = 0
    for t in range(1, n):
        good, _ = c.batch_matches(t)
        prev_kp = kd[t - 1][0][good["queryIdx"]]                      # Frame::nextGoodMatches of the previous frame
        l0, l1 = orc.half_pyramid(frames[t - 1]), orc.half_pyramid(frames[t])
        ogx, ogy = [], []
        for lv in l0:
            a, b, _ = orc.scharr_gradient(lv, 3)
            ogx.append(a); ogy.append(b)
        cand = [orc.patch_points(prev_kp, W, H, l) for l in range(5)]
        ref = orc.estimate_pose_features(orc.default_align_params(), W, H, l0, l1, ogx, ogy, cand)
        _same(res[t], ref)
        # and the single-pair entry point on the same host data
        _same(c.estimate_pose_features(ap, W, H, l0, l1, ogx, ogy, cand), ref)
        npairs_checked += 1
    assert npairs_checked == n - 1
    c.close()


def test_align_batch_explicit_points_and_init(vislam, orc, ctx, canvas):
    """vis_align_batch with caller-provided matched points (more than 200 per pair: only the first 200 are used, like
    min(num_max_keypoints, 200) at src/Camera.cpp:377) and per-pair initial poses"""
    import torch
    W, H, n = 320, 240, 3
    cv = vislam.synth_canvas(1024, 5)
    frames = np.stack([vislam.synth_frame(cv, t, W, H, 5) for t in range(n)])
    dev = torch.from_numpy(frames).cuda()
    fe = vislam.gradient_frame_elems(W, H)
    gray = torch.zeros(n * fe, dtype=torch.uint8, device="cuda")
    gx = torch.zeros(n * fe, dtype=torch.int16, device="cuda"); gy = torch.zeros_like(gx)
    g = torch.zeros(n * fe, dtype=torch.uint8, device="cuda")
    ctx.gradient_batch(dev.data_ptr(), W, H, W, n, gray.data_ptr(), gx.data_ptr(), gy.data_ptr(), g.data_ptr())
    rng = np.random.default_rng(0)
    max_pts = 230
    pts = np.zeros((n, max_pts, 2), np.float32)
    pts[..., 0] = rng.uniform(-2, W + 2, (n, max_pts)); pts[..., 1] = rng.uniform(-2, H + 2, (n, max_pts))   # some patches are clipped
    npts = np.array([0, 230, 17], np.int32)
    inits = np.zeros((n, 7), np.float32); inits[:, 3] = 1
    i2 = orc.se3_exp([0.001, 0, 0, 0, 0, 0.002]); inits[2] = i2.as_array()
    d_pts = torch.from_numpy(pts).cuda(); d_n = torch.from_numpy(npts).cuda(); d_init = torch.from_numpy(inits).cuda()
    out = torch.zeros(n * C.sizeof(vislam.AlignResult), dtype=torch.uint8, device="cuda")
    ap = vislam.default_align_params()
    ap.fx, ap.fy, ap.cx, ap.cy = 200.0, 200.0, 160.0, 120.0
    ctx.align_batch(ap, dev.data_ptr(), W, H, W, n, gray.data_ptr(), gx.data_ptr(), gy.data_ptr(), d_pts.data_ptr(), d_n.data_ptr(), max_pts,
                    d_init.data_ptr(), out.data_ptr())
    torch.cuda.synchronize()
    raw = out.cpu().numpy().tobytes()
    res = [vislam.AlignResult.from_buffer_copy(raw, i * C.sizeof(vislam.AlignResult)) for i in range(n)]
    KP = vislam.KEYPOINT_DTYPE
    oap = orc.default_align_params()
    oap.fx, oap.fy, oap.cx, oap.cy = 200.0, 200.0, 160.0, 120.0
    for t in (1, 2):
        kp = np.zeros(npts[t], KP); kp["x"], kp["y"] = pts[t, :npts[t], 0], pts[t, :npts[t], 1]
        l0, l1 = orc.half_pyramid(frames[t - 1]), orc.half_pyramid(frames[t])
        ogx, ogy = [], []
        for lv in l0:
            a, b, _ = orc.scharr_gradient(lv, 3)
            ogx.append(a); ogy.append(b)
        cand = [orc.patch_points(kp, W, H, l) for l in range(5)]
        init = None if t == 1 else i2
        _same(res[t], orc.estimate_pose_features(oap, W, H, l0, l1, ogx, ogy, cand, init))


def test_batch_align_is_not_overtaken_by_the_next_step(vislam, canvas):
    """vis_batch_align runs on the pose stream beside the next batch's detect chain.  The next step rewrites what it reads
    -- the plan's matched points (filter), the caller's gradient buffers (vis_gradient_batch into the SAME buffers) --
    so a pipelined run (no sync between the steps) must give every step exactly the results of a step-by-step run."""
    import torch
    W, H, n = 752, 480, 48
    p = vislam.default_params()
    c = vislam.Context(0, p)
    sets = [np.stack([vislam.synth_frame(canvas, 100 * s + 7 * t, W, H) for t in range(n)]) for s in range(3)]
    devs = [torch.from_numpy(f).cuda() for f in sets]
    c.batch_plan(W, H, W, n)
    fe = vislam.gradient_frame_elems(W, H)
    gray = torch.zeros(n * fe, dtype=torch.uint8, device="cuda")
    gx = torch.zeros(n * fe, dtype=torch.int16, device="cuda"); gy = torch.zeros_like(gx)
    g = torch.zeros(n * fe, dtype=torch.uint8, device="cuda")
    ap = vislam.default_align_params()
    sz = n * C.sizeof(vislam.AlignResult)

    def step(d, out):
        c.batch_run(d.data_ptr(), n, vislam.STAGE_DETECT | vislam.STAGE_MATCH)
        c.gradient_batch(d.data_ptr(), W, H, W, n, gray.data_ptr(), gx.data_ptr(), gy.data_ptr(), g.data_ptr())
        c.batch_align(ap, d.data_ptr(), n, gray.data_ptr(), gx.data_ptr(), gy.data_ptr(), 0, out.data_ptr())

    ref = []
    c.batch_reset()
    for d in devs:                                                    # step by step
        out = torch.zeros(sz, dtype=torch.uint8, device="cuda")
        step(d, out); c.batch_sync(); torch.cuda.synchronize()
        ref.append(out.cpu().numpy().copy())
    for rep in range(3):                                              # pipelined, a few times (a missing wait is a race)
        c.batch_reset()
        outs = [torch.zeros(sz, dtype=torch.uint8, device="cuda") for _ in devs]
        for d, out in zip(devs, outs):
            step(d, out)
        c.batch_sync(); torch.cuda.synchronize()
        for k, (out, r) in enumerate(zip(outs, ref)):
            assert np.array_equal(out.cpu().numpy(), r), (rep, k)
    assert any(r.any() for r in ref)
    c.close()


def test_stage_gradient_matches_gradient_batch_and_feeds_the_alignment(vislam, canvas):
    """VIS_STAGE_GRADIENT (plan-owned gradients on the side stream of vis_batch_run) = vis_gradient_batch on the same frames,
    and vis_batch_align with NULL gradient pointers = the same call with explicit buffers; pipelined steps included."""
    import torch
    W, H, n = 752, 480, 24
    c = vislam.Context(0, vislam.default_params())
    sets = [np.stack([vislam.synth_frame(canvas, 50 * s + 5 * t, W, H) for t in range(n)]) for s in range(2)]
    devs = [torch.from_numpy(f).cuda() for f in sets]
    c.batch_plan(W, H, W, n)
    fe = vislam.gradient_frame_elems(W, H)
    gray = torch.zeros(n * fe, dtype=torch.uint8, device="cuda")
    gx = torch.zeros(n * fe, dtype=torch.int16, device="cuda"); gy = torch.zeros_like(gx)
    g = torch.zeros(n * fe, dtype=torch.uint8, device="cuda")
    ap = vislam.default_align_params()
    sz = n * C.sizeof(vislam.AlignResult)
    ref = []
    for d in devs:                                                    # explicit buffers, step by step
        out = torch.zeros(sz, dtype=torch.uint8, device="cuda")
        c.batch_run(d.data_ptr(), n, vislam.STAGE_DETECT | vislam.STAGE_MATCH)
        c.gradient_batch(d.data_ptr(), W, H, W, n, gray.data_ptr(), gx.data_ptr(), gy.data_ptr(), g.data_ptr())
        c.batch_align(ap, d.data_ptr(), n, gray.data_ptr(), gx.data_ptr(), gy.data_ptr(), 0, out.data_ptr())
        c.batch_sync(); torch.cuda.synchronize()
        ref.append((out.cpu().numpy().copy(), gray.cpu().numpy().copy(), gx.cpu().numpy().copy(), gy.cpu().numpy().copy(), g.cpu().numpy().copy()))
    c.batch_reset()
    with pytest.raises(Exception):                                    # no gradients given, none in the plan
        c.batch_run(devs[0].data_ptr(), n, vislam.STAGE_DETECT | vislam.STAGE_MATCH)
        c.batch_align(ap, devs[0].data_ptr(), n, 0, 0, 0, 0, torch.zeros(sz, dtype=torch.uint8, device="cuda").data_ptr())
    c.batch_sync()
    c.batch_reset()
    outs = [torch.zeros(sz, dtype=torch.uint8, device="cuda") for _ in devs]
    for k, (d, out) in enumerate(zip(devs, outs)):                    # plan-owned gradients, pipelined
        c.batch_run(d.data_ptr(), n, vislam.STAGE_DETECT | vislam.STAGE_MATCH | vislam.STAGE_GRADIENT)
        c.batch_align(ap, d.data_ptr(), n, 0, 0, 0, 0, out.data_ptr())
    c.batch_sync(); torch.cuda.synchronize()
    pg, px, py, pgg, pfe = c.batch_gradients()
    assert pfe == fe
    for out, r in zip(outs, ref):
        assert np.array_equal(out.cpu().numpy(), r[0])
    # the plan's buffers hold the LAST batch's gradients
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    def fetch(ptr, nbytes):
        host = np.empty(nbytes, np.uint8)
        assert hip.hipMemcpy(ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), 2) == 0     # D2H
        return host
    last = ref[-1]
    lv0 = W * H
    got_gray = fetch(pg, n * fe).reshape(n, fe)[:, lv0:]
    assert np.array_equal(got_gray, last[1].reshape(n, fe)[:, lv0:])                          # levels 1..4 (level 0 is the frame)
    assert np.array_equal(fetch(px, n * fe * 2).view(np.int16), last[2])
    assert np.array_equal(fetch(py, n * fe * 2).view(np.int16), last[3])
    assert np.array_equal(fetch(pgg, n * fe), last[4])
    c.close()


def test_warped_points_in_the_extra_row_and_column(vislam, orc, ctx, canvas):
    """150 x 110: level 2 is a 38 x 28 Mat (cvRound(37.5), cvRound(27.5)) where the reference's bookkeeping says 37 x 27.  The reference
    tests a WARPED point against the Mat (`y2 < image2.rows && x2 < image2.cols`, src/VISystem.cpp:1299), so points that land in
    column 37.x / row 27.x count as residuals (round 4 rejected them with the bookkeeping size: ADVICE r4).  Hand-placed candidates
    at the right / bottom edge of level 2 and a pose that moves them by half a pixel: oracle and kernel agree bit for bit AND the
    residual count shows the points were accepted."""
    w, h = 150, 110
    c = align_cases.case(vislam, orc, canvas, w=w, h=h, dx=1, dy=1, n=20, grad_div=8)
    lw, lh = vislam.half_pyramid_dims(w, h)
    assert (lw[2], lh[2]) == (38, 28) and (w >> 2, h >> 2) == (37, 27)
    pts = np.array([[36.0, y, 1.0, 1.0] for y in range(4, 26, 3)] + [[x, 26.0, 1.0, 1.0] for x in range(4, 36, 3)], np.float32)
    cand = [np.zeros((0, 4), np.float32)] * 5
    cand[2] = pts
    ap = vislam.default_align_params(); oap = orc.default_align_params()
    for q in (ap, oap):
        q.fx, q.fy, q.cx, q.cy = 120.0, 120.0, w / 2.0, h / 2.0
        q.first_level, q.last_level, q.max_iterations = 2, 2, 1
    # level-2 focal length = 120 / 4 = 30: a translation of 0.04 along x and y moves every point (z = 1) by 1.2 px: x 36 -> 37.2, y 26 -> 27.2
    init = orc.se3_exp([0.04, 0.04, 0, 0, 0, 0])
    got = ctx.estimate_pose_features(ap, w, h, c["gray1"], c["gray2"], c["gx"], c["gy"], cand, init)
    ref = orc.estimate_pose_features(oap, w, h, c["gray1"], c["gray2"], c["gx"], c["gy"], cand, init)
    _same(got, ref)
    assert ref.n_residuals[2] == len(pts), (list(ref.n_residuals), len(pts))      # every point counted, including the ones in the extra row / column
