"""GPU: the three-stream batch pipeline (detect | match | pose, double-buffered records, carried last frame, queued D2H of
results, alignment on the matcher's output) is deterministic and independent of how a stream is cut into batches.
This is the race check of the build (the reference has none): any missing event between the streams
shows up as a run-to-run or batching-dependent difference."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
W, H = 752, 480


def _run(vislam, frames_dev, n_total, batch, with_align):
    import torch
    p = vislam.default_params()
    p.fy = p.fx
    c = vislam.Context(0, p)
    c.batch_plan(W, H, W, batch)
    root2 = 49
    poses, goods, ngs, kps, aligns = [], [], [], [], []
    fe = vislam.gradient_frame_elems(W, H)
    if with_align:
        gray = torch.zeros(batch * fe, dtype=torch.uint8, device="cuda")
        gx = torch.zeros(batch * fe, dtype=torch.int16, device="cuda"); gy = torch.zeros_like(gx)
        g = torch.zeros(batch * fe, dtype=torch.uint8, device="cuda")
        aout = torch.zeros(batch * C.sizeof(vislam.AlignResult), dtype=torch.uint8, device="cuda")
        ap = vislam.default_align_params()
    pending = None
    for b0 in range(0, n_total, batch):
        n = min(batch, n_total - b0)
        d = frames_dev.data_ptr() + b0 * W * H
        c.batch_run(d, n)
        hp = torch.zeros(n * C.sizeof(vislam.PoseResult), dtype=torch.uint8).pin_memory()
        hg = torch.zeros(n * root2 * 16, dtype=torch.uint8).pin_memory()
        hn = torch.zeros(n, dtype=torch.int32).pin_memory()
        c.batch_results_async(n, hp.data_ptr(), hg.data_ptr(), hn.data_ptr())
        if with_align:
            c.gradient_batch(d, W, H, W, n, gray.data_ptr(), gx.data_ptr(), gy.data_ptr(), g.data_ptr())
            c.batch_align(ap, d, n, gray.data_ptr(), gx.data_ptr(), gy.data_ptr(), 0, aout.data_ptr())
        if pending is not None and not with_align:
            pass                                               # the previous batch's download overlaps this batch: no sync in between
        if with_align:                                         # the alignment output buffer is reused by the next batch
            c.batch_sync()
            aligns.append(aout.cpu().numpy()[:n * C.sizeof(vislam.AlignResult)].copy())
        pending = (hp, hg, hn, n)
        # keypoints of this batch (synchronises: also exercises sync between batches in one of the two modes)
        if b0 % (2 * batch) == 0:
            kps.append(c.batch_keypoints(n - 1)[0].tobytes())
        else:
            kps.append(None)
        # results must be complete once the NEXT batch has been queued and synced; collect at the end of the loop body
        c.batch_sync()
        poses.append(np.frombuffer(hp.numpy().tobytes(), vislam.POSE_RESULT_DTYPE).copy())
        goods.append(np.frombuffer(hg.numpy().tobytes(), vislam.DMATCH_DTYPE).reshape(n, root2).copy())
        ngs.append(hn.numpy().copy())
    assert c.batch_status() == 0
    c.close()
    return np.concatenate(poses), np.concatenate(goods), np.concatenate(ngs), kps, (np.concatenate(aligns) if with_align else None)


def test_batching_independent_and_repeatable(vislam, canvas):
    import torch
    n_total = 96
    frames = np.stack([vislam.synth_frame(canvas, t, W, H, parallax=True) for t in range(n_total)])
    dev = torch.from_numpy(frames).cuda()
    ref = _run(vislam, dev, n_total, 96, False)
    for batch in (16, 32, 96):
        got = _run(vislam, dev, n_total, batch, False)
        assert got[0].tobytes() == ref[0].tobytes(), ("pose records differ", batch)
        assert got[2].tobytes() == ref[2].tobytes(), ("good match counts differ", batch)
        for t in range(n_total):                                 # rows are dense up to the count; the rest of a row is not written
            assert got[1][t, :ref[2][t]].tobytes() == ref[1][t, :ref[2][t]].tobytes(), ("good matches differ", batch, t)
    # pair 0 of the stream has no predecessor; every later frame has a pose record with correspondences
    assert ref[0]["n_points"][0] == 0 and (ref[0]["n_points"][1:] > 5).all()


def test_alignment_in_the_pipeline_is_repeatable(vislam, canvas):
    import torch
    n_total = 48
    frames = np.stack([vislam.synth_frame(canvas, t, W, H) for t in range(n_total)])
    dev = torch.from_numpy(frames).cuda()
    a = _run(vislam, dev, n_total, 24, True)
    b = _run(vislam, dev, n_total, 24, True)
    assert a[4].tobytes() == b[4].tobytes() and a[0].tobytes() == b[0].tobytes()
    res = np.frombuffer(a[4].tobytes(), np.uint8).reshape(n_total, -1)
    nres = np.frombuffer(res[:, 136:156].tobytes(), np.int32).reshape(n_total, 5)
    assert (nres[[0, 24]] == 0).all()                          # first frame of each batch: no predecessor inside the batch
    assert (nres[1:24, 0] > 1000).all() and (nres[25:, 0] > 1000).all()


def test_sync_and_timings_before_the_first_run_leave_no_error_behind(vislam, canvas):
    """vis_batch_sync / vis_timings query the stage events; before the first run (or after a run that left stages out) some of
    them were never recorded and the query fails -- that failure must not survive as the thread's HIP "last error", where the
    launch checks of the next, valid, call would pick it up (seen as `vis_batch_run: hipGetLastError(): invalid resource handle`
    from a bench leg with zero warm-up steps)."""
    import torch
    n = 8
    frames = np.stack([vislam.synth_frame(canvas, t, W, H) for t in range(n)])
    dev = torch.from_numpy(frames).cuda()
    p = vislam.default_params(); p.fy = p.fx
    ref = vislam.Context(0, p); ref.batch_plan(W, H, W, n)
    ref.batch_run(dev.data_ptr(), n); ref.batch_sync()
    want = ref.batch_results(n)[0].tobytes()
    ref.close()
    c = vislam.Context(0, p)
    c.batch_plan(W, H, W, n)
    c.batch_sync()                                              # nothing has run: every event query fails
    assert c.batch_status() == 0
    c.timings()
    c.batch_run(dev.data_ptr(), n, vislam.STAGE_DETECT)         # the match / pose events stay unrecorded
    c.batch_sync(); c.timings()
    c2 = vislam.Context(0, p); c2.batch_plan(W, H, W, n)        # a fresh context on the same thread: same stream of "last errors"
    c2.batch_run(dev.data_ptr(), n); c2.batch_sync()
    assert c2.batch_results(n)[0].tobytes() == want
    c.close(); c2.close()


def test_work_list_item_counter_is_reset_when_pair_zero_has_no_model(vislam, canvas):
    """k_hyp_roots_packed claims work-list items from a counter that the scan building the list zeroes.  The first pair of a fresh
    stream has no predecessor (no model to estimate: the scan leaves early for it) -- the counter must be zeroed all the same, or
    the second stream on a context skips every item beyond the resident grid and scores the previous stream's roots.  Two different
    streams through ONE context (reset in between), RANSAC with the adaptive stop off and 2000 iterations so that the list is
    long (159 pairs x 32 items); the second must equal the same stream on a fresh context."""
    import torch
    n = 160
    p = vislam.default_params(); p.fy = p.fx
    p.ransac_adaptive, p.ransac_max_iters = 0, 2000
    fa = np.stack([vislam.synth_frame(canvas, t, W, H) for t in range(n)])
    fb = np.stack([vislam.synth_frame(canvas, 500 + 2 * t, W, H) for t in range(n)])
    da, db = torch.from_numpy(fa).cuda(), torch.from_numpy(fb).cuda()

    def run(ctx, d):
        ctx.batch_run(d.data_ptr(), n); ctx.batch_sync()
        assert ctx.batch_status() == 0
        pose, good, ng = ctx.batch_results(n)
        return pose.tobytes(), ng.tobytes()

    fresh = vislam.Context(0, p); fresh.batch_plan(W, H, W, n)
    want = run(fresh, db)
    fresh.close()
    c = vislam.Context(0, p); c.batch_plan(W, H, W, n)
    run(c, da)
    c.batch_reset()                                             # a new stream: pair 0 has no predecessor again
    got = run(c, db)
    c.close()
    assert got[1] == want[1] and got[0] == want[0]


def test_detect_chain_running_ahead_of_a_slow_matcher(vislam, canvas):
    """Since round 5 the detect chain of step i + 2 waits for the record set that step i's matcher reads only in front of k_describe
    (resize, FAST and select run ahead).  With 752x480 / 1000 keypoints the matcher is the faster stage and the wait rarely blocks; here
    it is the slow one (640x480 frames at FAST threshold 7 with a quota of 12000: 8600 keypoints per frame, two 8600 x 8600 distance
    matrices per pair -- 0.47 ms of k_knn_mfma + k_filter per step of 16 frames against 0.35 ms for the whole detect chain, measured),
    steps are queued back to back without a host sync, and the results must equal the same stream run one step at a
    time with a sync after every call (nothing overlaps there)."""
    import torch
    w, h, n_total, batch = 640, 480, 96, 16
    frames = np.stack([vislam.synth_frame(canvas, t, w, h, parallax=True) for t in range(n_total)])
    dev = torch.from_numpy(frames).cuda()
    p = vislam.default_params(); p.fy = p.fx
    p.nfeatures, p.w_size, p.h_size, p.fast_threshold = 12000, w, h, 7
    root2 = int(np.floor(np.sqrt(p.n_cells))) ** 2

    def run(serial):
        c = vislam.Context(0, p)
        c.batch_plan(w, h, w, batch)
        bufs = []
        for b0 in range(0, n_total, batch):
            c.batch_run(dev.data_ptr() + b0 * w * h, batch)
            hp = torch.zeros(batch * C.sizeof(vislam.PoseResult), dtype=torch.uint8).pin_memory()
            hg = torch.zeros(batch * root2 * 16, dtype=torch.uint8).pin_memory()
            hn = torch.zeros(batch, dtype=torch.int32).pin_memory()
            c.batch_results_async(batch, hp.data_ptr(), hg.data_ptr(), hn.data_ptr())
            if serial:
                c.batch_sync()
            bufs.append((hp, hg, hn))
        c.batch_sync()
        assert c.batch_status() == 0
        kp_last = c.batch_keypoints(batch - 1)
        c.close()
        pose = np.concatenate([np.frombuffer(b[0].numpy().tobytes(), vislam.POSE_RESULT_DTYPE) for b in bufs])
        ng = np.concatenate([b[2].numpy() for b in bufs])
        good = np.concatenate([np.frombuffer(b[1].numpy().tobytes(), vislam.DMATCH_DTYPE).reshape(batch, root2) for b in bufs])
        return pose, ng, good, kp_last

    ref = run(True)
    got = run(False)
    assert got[0].tobytes() == ref[0].tobytes() and got[1].tobytes() == ref[1].tobytes()
    for t in range(n_total):
        assert got[2][t, :ref[1][t]].tobytes() == ref[2][t, :ref[1][t]].tobytes(), t
    assert got[3][0].tobytes() == ref[3][0].tobytes() and got[3][1].tobytes() == ref[3][1].tobytes()
    assert (ref[0]["n_points"][1:] > 5).all()


def test_results_download_pinned_kernel_path_equals_runtime_copy_path(vislam, canvas):
    """vis_batch_results_async writes pinned (device-accessible) destinations with one kernel of the library and everything else with
    hipMemcpyAsync: the same batch through both must give the same bytes (pose records, counts, the dense part of the match rows)."""
    import torch
    n = 48
    frames = np.stack([vislam.synth_frame(canvas, t, W, H, parallax=True) for t in range(n)])
    dev = torch.from_numpy(frames).cuda()
    p = vislam.default_params(); p.fy = p.fx
    c = vislam.Context(0, p); c.batch_plan(W, H, W, n)
    root2 = 49
    c.batch_run(dev.data_ptr(), n)
    hp = torch.zeros(n * C.sizeof(vislam.PoseResult), dtype=torch.uint8).pin_memory()
    hg = torch.zeros(n * root2 * 16, dtype=torch.uint8).pin_memory()
    hn = torch.zeros(n, dtype=torch.int32).pin_memory()
    c.batch_results_async(n, hp.data_ptr(), hg.data_ptr(), hn.data_ptr())          # pinned: the kernel path
    c.batch_sync()
    pose, good, ng = c.batch_results(n)                                             # pageable numpy arrays: the runtime's copies
    c.close()
    assert hp.numpy().tobytes() == pose.tobytes() and hn.numpy().tobytes() == ng.tobytes()
    g1 = np.frombuffer(hg.numpy().tobytes(), vislam.DMATCH_DTYPE).reshape(n, root2)
    for t in range(n):
        assert g1[t, :ng[t]].tobytes() == good[t, :ng[t]].tobytes(), t
    assert (ng[1:] > 5).all()
