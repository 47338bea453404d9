"""Camera::computeGradient and the patch builders (the step after matching inside
CameraGPU::addGPUKeyframe, /root/reference/src/CameraGPU.cpp:154-157) -- HIP path vs the CPU oracle, bit exact
(integer stage; the point lists are float conversions of integers / of one double expression)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check_levels(vislam, orc, ctx, img, scale=3):
    gx, gy, g = ctx.compute_gradient(img, scale)
    gray = orc.half_pyramid(img)
    for l in range(5):
        ox, oy, og = orc.scharr_gradient(gray[l], scale)
        assert gx[l].shape == ox.shape
        assert (gx[l] == ox).all(), ("gx", l)
        assert (gy[l] == oy).all(), ("gy", l)
        assert (g[l] == og).all(), ("g", l)


def test_gradient_s752(vislam, orc, ctx, canvas):
    _check_levels(vislam, orc, ctx, vislam.synth_frame(canvas, 5, 752, 480))


@pytest.mark.parametrize("w,h", [(64, 48), (96, 32), (160, 112), (1920, 1088), (3840, 2160),
                                 (1920, 1080), (137, 135), (150, 110), (333, 61), (160, 118), (64, 50), (48, 34)])       # sizes that do not halve exactly: 1080 -> .. 135 -> 68 rows; width % 16 == 0 with any height: k_half_all + k_half4 for the rows below the last complete block row
def test_gradient_sizes(vislam, orc, ctx, canvas, w, h):
    # 64x48: level 4 is 4x3 (narrower than one thread's 8 pixels); 1920x1088: levels of every alignment class
    if w > 2048:        # the 4K configuration: textured random image (the 4096^2 fixture canvas is too small to crop it at t = 3)
        img = np.random.default_rng(21).integers(0, 256, (h, w), dtype=np.uint8)
    else:
        img = vislam.synth_frame(canvas, 3, w, h)
    _check_levels(vislam, orc, ctx, img)


@pytest.mark.parametrize("scale", [1, 3, 8])
def test_gradient_scale_and_extremes(vislam, orc, ctx, scale):
    # black/white checker blocks: the largest Scharr responses (16 * 255 * scale must not wrap in int16)
    yy, xx = np.mgrid[0:96, 0:128]
    img = (((xx // 3 + yy // 5) & 1) * 255).astype(np.uint8)
    _check_levels(vislam, orc, ctx, img, scale)
    rng = np.random.default_rng(7)
    _check_levels(vislam, orc, ctx, rng.integers(0, 256, (80, 144), dtype=np.uint8), scale)


def test_gradient_transpose_property_full_size(vislam, ctx):
    """size-independent property at the 4K configuration: the Scharr pair and the blend are symmetric under transposition
    (gx of the transposed image is gy transposed), level by level -- the half pyramid's 2x2 box mean is symmetric too"""
    img = np.random.default_rng(22).integers(0, 256, (2144, 2144), dtype=np.uint8)   # 2144 = 16 * 134
    gx, gy, g = ctx.compute_gradient(img)
    tx, ty, tg = ctx.compute_gradient(np.ascontiguousarray(img.T))
    for l in range(5):
        assert (tx[l] == gy[l].T).all() and (ty[l] == gx[l].T).all() and (tg[l] == g[l].T).all(), l


def test_gradient_invalid_arguments(vislam, ctx):
    img = np.zeros((48, 64), np.uint8)
    with pytest.raises(vislam.VisError):
        ctx.compute_gradient(img, scale=9)           # would overflow int16
    with pytest.raises(vislam.VisError):
        ctx.compute_gradient(np.zeros((12, 64), np.uint8))   # fewer than 16 rows


def test_gradient_batch_device_path(vislam, orc, ctx, canvas):
    """the batched device entry point (what the throughput path calls) on a strided frame buffer"""
    import torch
    W, H, n = 752, 480, 5
    frames = np.stack([vislam.synth_frame(canvas, t, W, H) for t in range(n)])
    d = torch.from_numpy(frames).cuda()
    fe = vislam.gradient_frame_elems(W, H)
    assert fe % 64 == 0 and fe >= sum((W >> l) * (H >> l) for l in range(5))
    gray = torch.zeros(n * fe, dtype=torch.uint8, device="cuda")
    gx = torch.zeros(n * fe, dtype=torch.int16, device="cuda"); gy = torch.zeros_like(gx)
    g = torch.zeros(n * fe, dtype=torch.uint8, device="cuda")
    ctx.gradient_batch(d.data_ptr(), W, H, W, n, gray.data_ptr(), gx.data_ptr(), gy.data_ptr(), g.data_ptr())
    torch.cuda.synchronize()                         # the call is asynchronous on the context's stream
    hx, hy, hg, hgray = gx.cpu().numpy(), gy.cpu().numpy(), g.cpu().numpy(), gray.cpu().numpy()
    for f in (0, 3, 4):
        ref = orc.half_pyramid(frames[f])
        off = f * fe
        for l in range(5):
            hl, wl = H >> l, W >> l
            ox, oy, og = orc.scharr_gradient(ref[l])
            sl = slice(off, off + hl * wl)
            assert (hx[sl].reshape(hl, wl) == ox).all(), (f, l)
            assert (hy[sl].reshape(hl, wl) == oy).all(), (f, l)
            assert (hg[sl].reshape(hl, wl) == og).all(), (f, l)
            if l:
                assert (hgray[sl].reshape(hl, wl) == ref[l]).all(), (f, l)
            off += hl * wl


def test_patch_and_debug_points(vislam, orc, ctx):
    rng = np.random.default_rng(11)
    p = vislam.default_params()
    p.w_size, p.h_size = 752, 480
    ctx.set_params(p)
    for n in (0, 1, 49, 230):                        # 230 > the reference's cap of 200 keypoints
        good = np.zeros(n, vislam.KEYPOINT_DTYPE)
        good["x"] = rng.uniform(-3, 755, n).astype(np.float32)      # includes points at / beyond the borders
        good["y"] = rng.uniform(-3, 483, n).astype(np.float32)
        patch, debug = ctx.patch_points(good)
        for l in range(5):
            ref = orc.patch_points(good, 752, 480, l)
            assert patch[l].shape == ref.shape, (n, l)
            assert (patch[l] == ref).all(), (n, l)
            dref = orc.debug_points(good, l)
            assert debug[l].shape == dref.shape and (debug[l] == dref).all(), (n, l)
    # capacity: a too small output reports the needed count
    good = np.zeros(10, vislam.KEYPOINT_DTYPE); good["x"] = 300; good["y"] = 200
    with pytest.raises(vislam.VisError):
        ctx.patch_points(good, cap=16)
