"""GPU: the threading contract of the C ABI (include/vislam_hip.h: a vis_ctx is not thread-safe, one context per host thread) -- two
host threads, each with its own context on the same device, run the frame-at-a-time entry points concurrently (ctypes releases the GIL
for the duration of a call); every result must equal the single-threaded one.  Process-wide state of the library is a launch counter
(atomic) and the kernels' function attributes."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
W, H = 752, 480


def _stream(vislam, canvas, t0, n):
    p = vislam.default_params(); p.fy = p.fx
    c = vislam.Context(0, p)
    out = []
    prev = None
    for i in range(n):
        img = vislam.synth_frame(canvas, t0 + i, W, H, parallax=True)
        c.camera_update(img)
        k, d = c.orb_detect_compute(img, slot=i & 1)
        rec = [k.tobytes(), d.tobytes()]
        if prev is not None:
            good, sym = c.good_matches((i - 1) & 1, i & 1)
            rec += [good.tobytes(), sym.tobytes()]
            if len(good) >= 5:
                p1 = np.array([[prev[g["queryIdx"]]["x"], prev[g["queryIdx"]]["y"]] for g in good], np.float32)
                p2 = np.array([[k[g["trainIdx"]]["x"], k[g["trainIdx"]]["y"]] for g in good], np.float32)
                E, mask, ninl, iters = c.essential_ransac(p1, p2)
                rec += [np.asarray(E).tobytes(), np.asarray(mask).tobytes(), int(ninl), int(iters)]
        prev = k
        out.append(rec)
    c.close()
    return out


def test_two_host_threads_two_contexts(vislam, canvas):
    n = 12
    want = [_stream(vislam, canvas, 0, n), _stream(vislam, canvas, 300, n)]
    got = [None, None]
    errs = []

    def work(j, t0):
        try:
            got[j] = _stream(vislam, canvas, t0, n)
        except Exception as e:                                      # noqa: BLE001 -- reported below
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(0, 0)), threading.Thread(target=work, args=(1, 300))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert got[0] == want[0] and got[1] == want[1]
