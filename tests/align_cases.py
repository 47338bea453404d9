"""Shared inputs of the alignment tests (VISystem::EstimatePoseFeatures): synthetic two-view frames
(the second view is the first shifted by whole pixels: a crop of one larger synthetic frame), their half pyramids,
Scharr gradients and the candidate point lists Camera::ObtainPatchesPointsPreviousFrame builds."""
import numpy as np


def two_frames(vislam, canvas, w, h, dx, dy, t=0, seed=0xE0C00001):
    big = vislam.synth_frame(canvas, t, w + 64, h + 64, seed)
    f0 = np.ascontiguousarray(big[32:32 + h, 32:32 + w])
    f1 = np.ascontiguousarray(big[32 - dy:32 - dy + h, 32 - dx:32 - dx + w])      # the scene moves by (+dx, +dy) pixels
    return f0, f1


def keypoints_of(vislam, orc, img, n=49, nfeatures=1000):
    p = vislam.default_params()
    p.nfeatures = nfeatures
    p.w_size, p.h_size = img.shape[1], img.shape[0]
    k, _ = orc.orb_detect_compute(p, img)
    step = max(1, len(k) // n)
    return k[::step][:n].copy()


def case(vislam, orc, canvas, w=752, h=480, dx=2, dy=1, n=49, grad_div=1, scale=3, t=0, seed=0xE0C00001):
    """-> dict(gray1, gray2, gx, gy, cand, kps): per-level lists.  grad_div > 1 divides the Scharr response (the
    reference's scale-3 Scharr is 96x the per-pixel intensity slope, which makes its Gauss-Newton steps tiny;
    grad_div = 96 gives the step size a textbook formulation would take and exercises many more iterations)."""
    f0, f1 = two_frames(vislam, canvas, w, h, dx, dy, t, seed)
    kps = keypoints_of(vislam, orc, f0, n)
    l0, l1 = orc.half_pyramid(f0), orc.half_pyramid(f1)
    gx, gy = [], []
    for lv in l0:
        a, b, _ = orc.scharr_gradient(lv, scale)
        if grad_div > 1:
            a = (a // grad_div).astype(np.int16); b = (b // grad_div).astype(np.int16)
        gx.append(a); gy.append(b)
    cand = [orc.patch_points(kps, w, h, l) for l in range(5)]
    return dict(gray1=l0, gray2=l1, gx=gx, gy=gy, cand=cand, kps=kps, f0=f0, f1=f1)


def result_tuple(r):
    return (list(r.iterations), list(r.n_residuals), [float(x) for x in r.error], float(r.initial_error),
            [float(x) for x in r.pose.as_array()], [float(x) for x in r.matrix])
