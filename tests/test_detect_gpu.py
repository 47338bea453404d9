"""GPU parity: ORB detect+describe through the C ABI vs the CPU oracle (bit-exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _params(vislam, n=1000, levels=8, w=752, h=480):
    p = vislam.default_params()
    p.nfeatures, p.nlevels, p.w_size, p.h_size = n, levels, w, h
    return p


def _assert_same(k, d, ok, od):
    assert len(k) == len(ok), (len(k), len(ok))
    for f in ("octave", "x", "y", "response", "angle", "size", "class_id"):
        bad = np.nonzero(k[f] != ok[f])[0]
        assert len(bad) == 0, (f, bad[:5], k[f][bad[:5]], ok[f][bad[:5]])
    assert k.tobytes() == ok.tobytes()
    diff = np.nonzero((d != od).any(axis=1))[0]
    assert len(diff) == 0, ("descriptor rows differ", diff[:10])


@pytest.mark.parametrize("t", [0, 1, 37])
def test_orb_752x480_bit_exact(vislam, orc, ctx, canvas, t):
    p = _params(vislam)
    ctx.set_params(p)
    img = vislam.synth_frame(canvas, t, 752, 480)
    k, d = ctx.orb_detect_compute(img, slot=0)
    ok, od = orc.orb_detect_compute(p, img)
    assert 900 <= len(ok) <= 1100
    _assert_same(k, d, ok, od)


@pytest.mark.parametrize("w,h,n,levels", [(320, 240, 300, 8), (641, 479, 500, 5), (188, 120, 200, 3), (1024, 768, 2000, 4)])
def test_orb_other_shapes(vislam, orc, ctx, canvas, w, h, n, levels):
    p = _params(vislam, n, levels, w, h)
    ctx.set_params(p)
    img = vislam.synth_frame(canvas, 3, w, h)
    k, d = ctx.orb_detect_compute(img, slot=1)
    ok, od = orc.orb_detect_compute(p, img)
    _assert_same(k, d, ok, od)


def test_orb_random_noise_image(vislam, orc, ctx):
    """dense corners everywhere: exercises ties at the FAST cut and big candidate lists"""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (240, 320), dtype=np.uint8)
    p = _params(vislam, 400, 6, 320, 240)
    ctx.set_params(p)
    k, d = ctx.orb_detect_compute(img, slot=0)
    ok, od = orc.orb_detect_compute(p, img)
    _assert_same(k, d, ok, od)


def test_orb_flat_image_no_keypoints(vislam, orc, ctx):
    p = _params(vislam, 500, 8, 320, 240)
    ctx.set_params(p)
    img = np.full((240, 320), 77, np.uint8)
    k, d = ctx.orb_detect_compute(img, slot=0)
    assert len(k) == 0 and len(d) == 0


def test_orb_strided_input(vislam, orc, ctx, canvas):
    p = _params(vislam, 500, 8, 400, 300)
    ctx.set_params(p)
    big = vislam.synth_frame(canvas, 9, 512, 300)
    view = big[:, 50:450]                 # non-contiguous rows (stride 512)
    img = np.ascontiguousarray(view)
    k, d = ctx.orb_detect_compute(img, slot=0)
    ok, od = orc.orb_detect_compute(p, img)
    _assert_same(k, d, ok, od)


def test_camera_update_half_pyramid(vislam, orc, ctx, canvas):
    img = vislam.synth_frame(canvas, 2, 752, 480)
    got = ctx.camera_update(img)
    ref = orc.half_pyramid(img)
    for l in range(5):
        assert got[l].shape == ref[l].shape
        assert (got[l] == ref[l]).all(), l


@pytest.mark.parametrize("w,h", [(1920, 1080), (137, 135), (150, 110), (61, 47), (19, 18)])
def test_camera_update_sizes_that_do_not_halve_exactly(vislam, orc, ctx, w, h):
    """cv::resize(Size(), 0.5, 0.5): level sizes cvRound(size * 0.5); partial 2x2 blocks average the pixels that exist (Camera.cpp:68-70)"""
    img = np.random.default_rng(w * 7 + h).integers(0, 256, (h, w), dtype=np.uint8)
    got, ref = ctx.camera_update(img), orc.half_pyramid(img)
    assert vislam.half_pyramid_dims(w, h) == orc.half_pyramid_dims(w, h)
    for l in range(5):
        assert got[l].shape == ref[l].shape and (got[l] == ref[l]).all(), l


def test_batch_update_and_gradient_stage_at_1080p(vislam, orc):
    """VIS_STAGE_UPDATE / VIS_STAGE_GRADIENT on 1920x1080 frames (BASELINE configs[2]): 1080 -> 540 -> 270 -> 135 -> 68 rows"""
    import torch
    W, H, n = 1920, 1080, 3
    p = vislam.default_params()
    p.nfeatures, p.nlevels, p.w_size, p.h_size = 4000, 4, W, H
    c = vislam.Context(0, p)
    cv = vislam.synth_canvas(4096, 0xE0C00003)
    frames = np.stack([vislam.synth_frame(cv, t, W, H, 0xE0C00003) for t in range(n)])
    dev = torch.from_numpy(frames).cuda()
    c.batch_plan(W, H, W, n)
    c.batch_run(dev.data_ptr(), n, vislam.STAGE_DETECT | vislam.STAGE_GRADIENT)
    c.batch_sync()
    assert c.batch_status() == 0
    lw, lh = vislam.half_pyramid_dims(W, H)
    assert lh == [1080, 540, 270, 135, 68]
    fe = vislam.gradient_frame_elems(W, H)
    pg, px, py, pgg, pfe = c.batch_gradients()
    assert pfe == fe

    def view(ptr, typestr):
        class _Dev:
            __cuda_array_interface__ = {"data": (ptr, False), "shape": (n * fe,), "typestr": typestr, "version": 2}
        return torch.as_tensor(_Dev(), device="cuda").cpu().numpy().reshape(n, fe)
    half, gx, gy, g = view(pg, "|u1"), view(px, "<i2"), view(py, "<i2"), view(pgg, "|u1")
    for t in range(n):
        ref = orc.half_pyramid(frames[t])
        off = 0
        for l in range(5):
            cnt = lw[l] * lh[l]
            ox, oy, og = orc.scharr_gradient(ref[l], 3)
            if l > 0:
                assert np.array_equal(half[t, off:off + cnt].reshape(lh[l], lw[l]), ref[l]), (t, l)
            assert np.array_equal(gx[t, off:off + cnt].reshape(lh[l], lw[l]), ox), (t, l)
            assert np.array_equal(gy[t, off:off + cnt].reshape(lh[l], lw[l]), oy), (t, l)
            assert np.array_equal(g[t, off:off + cnt].reshape(lh[l], lw[l]), og), (t, l)
            off += cnt
    c.close()


def test_batch_update_stage_writes_the_half_pyramids(vislam, orc, canvas):
    """VIS_STAGE_UPDATE (Camera::Update inside the batched step): every frame's 4 half levels == the oracle's, and the detect
    results of the same call are unchanged by it"""
    import torch
    p = _params(vislam)
    c = vislam.Context(0, p)
    n = 3
    frames = np.stack([vislam.synth_frame(canvas, 10 + t, 752, 480) for t in range(n)])
    dev = torch.from_numpy(frames).cuda()
    c.batch_plan(752, 480, 752, 4)
    with pytest.raises(vislam.VisError):                       # nothing written yet
        c.batch_half_pyramid()
    c.batch_run(dev.data_ptr(), n, vislam.STAGE_FRAME)
    c.batch_sync()
    assert c.batch_status() == 0
    assert c.timings().ms_update > 0
    ptr, fe = c.batch_half_pyramid()
    assert fe == vislam.gradient_frame_elems(752, 480)
    class _Dev:                                                  # the plan's buffer as a torch view (no copy API needed)
        __cuda_array_interface__ = {"data": (ptr, False), "shape": (n * fe,), "typestr": "|u1", "version": 2}
    buf = torch.as_tensor(_Dev(), device="cuda")
    half = buf.cpu().numpy().reshape(n, fe)
    for t in range(n):
        ref = orc.half_pyramid(frames[t])
        off = 752 * 480
        for l in range(1, 5):
            sz = ref[l].size
            assert np.array_equal(half[t, off:off + sz].reshape(ref[l].shape), ref[l]), (t, l)
            off += sz
        k, d = c.batch_keypoints(t)
        ok, od = orc.orb_detect_compute(p, frames[t])
        _assert_same(k, d, ok, od)
    c.batch_run(dev.data_ptr(), n, vislam.STAGE_ALL)            # without the stage the accessor refuses again
    c.batch_sync()
    with pytest.raises(vislam.VisError):
        c.batch_half_pyramid()
    c.close()


def test_speculative_fast_threshold_is_exact(vislam, orc, canvas):
    """batched streams run FAST at a per-level threshold predicted from the previous batch's retainBest cuts (detect.hip fast_tile);
    k_select verifies it per (frame, level) and a device work list redoes what the prediction got wrong.  Batches chosen so that
    the prediction is (1) absent, (2) far too high (noise -> stream: the fix-up path redoes everything), (3) right (stream ->
    stream), (4) too high again (stream -> low contrast), (5) too low / harmless (low contrast -> noise): every frame of every batch
    must equal the oracle's FAST-at-20 result bit for bit"""
    import torch
    p = _params(vislam)
    c = vislam.Context(0, p)
    rng = np.random.default_rng(5)
    n = 4
    stream = lambda t0: np.stack([vislam.synth_frame(canvas, t0 + t, 752, 480) for t in range(n)])      # noqa: E731
    noise = rng.integers(0, 256, (n, 480, 752), dtype=np.uint8)
    low = (stream(40).astype(np.int32) // 3 + 80).astype(np.uint8)                                           # a third of the contrast
    batches = [stream(0), noise, stream(8), stream(12), low, noise, stream(16)]
    c.batch_plan(752, 480, 752, n)
    redone, taus = [], []
    for bi, frames in enumerate(batches):
        dev = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
        c.batch_run(dev.data_ptr(), n, vislam.STAGE_DETECT)
        c.batch_sync()
        assert c.batch_status() == 0, bi
        tau, nr = c.batch_fast_thresholds()
        redone.append(nr); taus.append(tau.copy())
        for t in range(n):
            k, d = c.batch_keypoints(t)
            ok, od = orc.orb_detect_compute(p, frames[t])
            _assert_same(k, d, ok, od)
    # what the prediction machinery did: nothing to redo without a prediction (batch 0) or with a good one (stream -> stream),
    # every (frame, level) redone when the previous batch promised far more (noise -> stream, stream -> low contrast)
    assert redone[0] == 0 and redone[3] == 0, redone
    assert redone[1] > 0 and redone[2] > 0 and redone[4] == n * 8, redone
    assert (taus[2] > 40).all() and (taus[4] == 20).any(), taus          # after a stream batch / after a low-contrast batch
    # a reset forgets the prediction (new stream): still exact
    c.batch_reset()
    assert (c.batch_fast_thresholds()[0] == 20).all()
    dev = torch.from_numpy(low).cuda()
    c.batch_run(dev.data_ptr(), n, vislam.STAGE_DETECT)
    c.batch_sync()
    for t in range(n):
        k, d = c.batch_keypoints(t)
        ok, od = orc.orb_detect_compute(p, low[t])
        _assert_same(k, d, ok, od)
    c.close()


def test_batch_matches_single(vislam, orc, canvas):
    """batched device path == single-frame path == oracle, including the carried frame across batches"""
    import torch
    p = _params(vislam)
    c = vislam.Context(0, p)
    frames = np.stack([vislam.synth_frame(canvas, t, 752, 480) for t in range(5)])
    dev = torch.from_numpy(frames).cuda()
    c.batch_plan(752, 480, 752, 3)
    c.batch_run(dev.data_ptr(), 3)
    c.batch_sync()
    assert c.batch_status() == 0
    res = [c.batch_keypoints(i) for i in range(3)]
    c.batch_run(dev.data_ptr() + 3 * 752 * 480, 2)
    c.batch_sync()
    res += [c.batch_keypoints(i) for i in range(2)]
    prev = None
    for t in range(5):
        ok, od, r = orc.pipeline_frame(p, frames[t], prev)
        _assert_same(res[t][0], res[t][1], ok, od)
        prev = (ok, od)
    # pair results of the second batch: frame 3 is matched against the carried frame 2
    g, nsym = c.batch_matches(0)
    o12, o21 = orc.knn2_hamming(res[2][1], res[3][1])
    og, osym = orc.good_matches(p, res[2][0], res[3][0], o12, o21)
    assert nsym == len(osym) and g.tobytes() == og.tobytes()
    c.close()


@pytest.mark.parametrize("n", [21, 150])
def test_batch_frame_tails_and_keypoint_walk(vislam, orc, canvas, n):
    """batches whose size is not a multiple of 8 (the (xcd, item, frame/8) grids have idle workgroups) and, at n = 150,
    large enough that a describe wave walks over several keypoints (16384 / n workgroups per frame < kcap / 4): every frame
    of the batch against the oracle"""
    import torch
    p = _params(vislam, n=300, levels=6, w=320, h=240)
    c = vislam.Context(0, p)
    frames = np.stack([vislam.synth_frame(canvas, t, 320, 240) for t in range(n)])
    dev = torch.from_numpy(frames).cuda()
    c.batch_plan(320, 240, 320, n)
    c.batch_run(dev.data_ptr(), n, vislam.STAGE_DETECT)
    c.batch_sync()
    assert c.batch_status() == 0
    for t in range(n):
        k, d = c.batch_keypoints(t)
        ok, od = orc.orb_detect_compute(p, frames[t])
        _assert_same(k, d, ok, od)
    c.close()
