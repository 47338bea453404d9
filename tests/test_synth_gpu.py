"""GPU: the device-side stream generator writes the same bytes as the host generator (both modes), and the batched
results download (vis_batch_results_async) returns what the per-frame getters return."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("parallax", [False, True])
@pytest.mark.parametrize("w,h,dim,t0,n", [(752, 480, 2048, 0, 5), (321, 243, 1024, 1000, 3), (1920, 1080, 4096, 7, 2)])
def test_device_generator_matches_host(vislam, ctx, w, h, dim, t0, n, parallax):
    import torch
    seed = 0xE0C00010 + dim
    cv = vislam.synth_canvas(dim, seed)
    d_cv = torch.from_numpy(cv).cuda()
    stride = (w + 3) // 4 * 4 + 8
    out = torch.zeros((n, h, stride), dtype=torch.uint8, device="cuda")
    ctx.synth_frames_device(d_cv.data_ptr(), dim, seed, t0, n, w, h, stride, out.data_ptr(), parallax)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for i in range(n):
        ref = vislam.synth_frame(cv, t0 + i, w, h, seed, parallax=parallax)
        assert np.array_equal(got[i, :, :w], ref), (i, parallax)
        assert not got[i, :, w:].any()                                  # row padding untouched


def test_batch_results_download(vislam, orc, canvas):
    import torch
    p = vislam.default_params()
    p.fy = p.fx
    c = vislam.Context(0, p)
    n = 5
    frames = np.stack([vislam.synth_frame(canvas, t, 752, 480, parallax=True) for t in range(n)])
    dev = torch.from_numpy(frames).cuda()
    c.batch_plan(752, 480, 752, 8)
    c.batch_run(dev.data_ptr(), n)
    hp = torch.zeros(n * C.sizeof(vislam.PoseResult), dtype=torch.uint8).pin_memory()
    hg = torch.zeros(n * 49 * 16, dtype=torch.uint8).pin_memory()
    hn = torch.zeros(n, dtype=torch.int32).pin_memory()
    with pytest.raises(vislam.VisError):                               # buffers for fewer frames than the batch had: refused, nothing copied
        c.batch_results_async(n - 1, hp.data_ptr(), hg.data_ptr(), hn.data_ptr())
    c.batch_results_async(n, hp.data_ptr(), hg.data_ptr(), hn.data_ptr())
    c.batch_run(dev.data_ptr(), n)                                     # the next batch must not disturb the queued copy
    c.batch_sync()
    pose = np.frombuffer(hp.numpy().tobytes(), vislam.POSE_RESULT_DTYPE)
    good = np.frombuffer(hg.numpy().tobytes(), vislam.DMATCH_DTYPE).reshape(n, 49)
    # reference: the same frames through a fresh stream, per-frame getters + the oracle
    d = vislam.Context(0, p)
    d.batch_plan(752, 480, 752, 8)
    d.batch_run(dev.data_ptr(), n)
    d.batch_sync()
    prev = None
    for t in range(n):
        g, nsym = d.batch_matches(t)
        ps = d.batch_pose(t)
        assert hn[t].item() == len(g) and good[t, :len(g)].tobytes() == g.tobytes()
        assert pose["n_inliers"][t] == ps["n_inliers"] and pose["iters_run"][t] == ps["iters_run"]
        assert np.array_equal(pose["E"][t].reshape(3, 3), ps["E"]) and np.array_equal(pose["R"][t].reshape(3, 3), ps["R"])
        ok, od, r = orc.pipeline_frame(p, frames[t], prev)
        prev = (ok, od)
        assert pose["n_points"][t] == r.n_good and pose["iters_run"][t] == r.iters_run and pose["n_inliers"][t] == r.n_inliers
        if t > 0:
            assert pose["n_models"][t] >= pose["iters_run"][t] > 1      # S-752P: RANSAC really iterates
    c.close(); d.close()
