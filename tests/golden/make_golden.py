#!/usr/bin/env python3
"""Regenerates the golden fixtures under tests/golden/.

The reference (MecatronicaUSB/vi-slam) ships no tests, fixtures or golden vectors, and OpenCV 3.2 is
not installed here, so these vectors are produced by THIS repo's CPU oracle (oracle/) on inputs from
the integer-only synthetic generator: they pin the oracle against regressions and give the GPU tests a
second, committed reference.  They do NOT pin parity with real OpenCV ("parity unpinned").
Fixtures are data only (inputs + expected outputs)."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vislam  # noqa: E402
import oracle_bind as orc  # noqa: E402


def main():
    cv = vislam.synth_canvas(512, 123)
    np.savez_compressed(os.path.join(HERE, "synth_512_123.npz"), canvas_sha256=hashlib.sha256(cv.tobytes()).hexdigest(),
                        frame5=vislam.synth_frame(cv, 5, 160, 120, 123))
    # ORB on two 160x120 frames, 3 levels, 120 features
    p = vislam.default_params()
    p.nfeatures, p.nlevels, p.w_size, p.h_size, p.n_cells = 120, 3, 160, 120, 16
    f0 = vislam.synth_frame(cv, 0, 160, 120, 123)
    f1 = vislam.synth_frame(cv, 1, 160, 120, 123)
    k0, d0 = orc.orb_detect_compute(p, f0)
    k1, d1 = orc.orb_detect_compute(p, f1)
    o12, o21 = orc.knn2_hamming(d0, d1)
    good, sym = orc.good_matches(p, k0, k1, o12, o21)
    p.sym_mode = 1
    good_i, sym_i = orc.good_matches(p, k0, k1, o12, o21)
    xs, ys, sc, smap = orc.fast_detect(f0, 20)
    lv1 = orc.resize_linear(f0, 133, 100)
    np.savez_compressed(os.path.join(HERE, "orb_160x120.npz"), f0=f0, f1=f1, k0=k0, d0=d0, k1=k1, d1=d1, knn12=o12, knn21=o21,
                        good=good, sym=sym, good_intended=good_i, sym_intended=sym_i, fast_xs=xs, fast_ys=ys, fast_sc=sc,
                        level1=lv1, half=np.concatenate([l.ravel() for l in orc.half_pyramid(f0[:112, :])[1:]]))
    # pose: synthetic two-view correspondences with a known R,t (float32 pixels)
    import test_pose_gpu
    x1, x2, R, t = test_pose_gpu.two_view(80, 99, 0.25, 0.3)
    q = vislam.default_params()
    q.fy = q.fx
    E, mask, ninl, iters = orc.essential_ransac(q, x1, x2)
    Rr, tr, ng = orc.recover_pose(q, E, x1, x2)
    rng = np.random.default_rng(4)
    idx = rng.integers(0, 79, (300, 2)).astype(np.int32)
    KP = vislam.KEYPOINT_DTYPE
    a, b = np.zeros(80, KP), np.zeros(80, KP)
    a["x"], a["y"], b["x"], b["y"] = x1[:, 0], x1[:, 1], x2[:, 0], x2[:, 1]
    ft, fc = orc.f2f_ransac(q, a, b, R.T.astype(np.float32), idx, 0.5)
    np.savez_compressed(os.path.join(HERE, "pose_80.npz"), x1=x1, x2=x2, R_true=R, t_true=t, E=E, mask=mask, ninl=ninl, iters=iters,
                        R=Rr, t=tr, ngood=ng, f2f_idx=idx, f2f_rot=R.T.astype(np.float32), f2f_t=ft, f2f_count=fc,
                        samples49=orc.ransac_samples(0xFFFFFFFFFFFFFFFF, 49, 20))
    # Camera::computeGradient (Scharr scale 3) on the 5 half-pyramid levels of a 160x112 crop + the patch point lists of
    # a hand-made keypoint set
    crop = np.ascontiguousarray(f0[:112, :])
    gxs, gys, gs = [], [], []
    for lv in orc.half_pyramid(crop):
        ox, oy, og = orc.scharr_gradient(lv, 3)
        gxs.append(ox.ravel()); gys.append(oy.ravel()); gs.append(og.ravel())
    kp = np.zeros(6, vislam.KEYPOINT_DTYPE)
    kp["x"] = [80.0, 3.25, 158.5, 41.75, 0.0, 120.0]
    kp["y"] = [56.0, 2.5, 110.0, 77.125, 0.0, 9.5]
    pts = [orc.patch_points(kp, 160, 112, l) for l in range(5)]
    dbg = [orc.debug_points(kp, l) for l in range(5)]
    np.savez_compressed(os.path.join(HERE, "gradient_160x112.npz"), img=crop, gx=np.concatenate(gxs), gy=np.concatenate(gys), g=np.concatenate(gs),
                        kp=kp, patch=np.concatenate(pts), patch_counts=np.array([len(x) for x in pts]), debug=np.concatenate(dbg))
    # VISystem::EstimatePoseFeatures (Gauss-Newton photometric alignment) on a 320x240 two-view case: second view = the
    # first shifted by (4, 3) px, 12 keypoints, gradients divided by 8 so that several iterations run per
    # level (7, 4, 1, 1): with the reference's scale-3 Scharr the steps are so small that most levels stop at k = 1
    import align_cases
    cv2 = vislam.synth_canvas(1024, 77)
    c = align_cases.case(vislam, orc, cv2, w=320, h=240, dx=4, dy=3, n=12, grad_div=8, seed=77)
    ap = orc.default_align_params()
    ap.fx, ap.fy, ap.cx, ap.cy = 200.0, 200.0, 160.0, 120.0
    r = orc.estimate_pose_features(ap, 320, 240, c["gray1"], c["gray2"], c["gx"], c["gy"], c["cand"])
    d = {}
    for l in range(5):
        d[f"gray1_{l}"] = c["gray1"][l]; d[f"gray2_{l}"] = c["gray2"][l]; d[f"gx_{l}"] = c["gx"][l]; d[f"gy_{l}"] = c["gy"][l]
        d[f"cand_{l}"] = c["cand"][l]
    np.savez_compressed(os.path.join(HERE, "align_320x240.npz"), iterations=np.array(list(r.iterations)), n_residuals=np.array(list(r.n_residuals)),
                        error=np.array(r.error, np.float32), pose=r.pose.as_array(), matrix=np.array(r.matrix, np.float32), **d)
    half_odd(cv)
    print("golden fixtures written:", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


def half_odd(cv):
    """Camera::Update + Camera::computeGradient on a size that does not halve exactly (150 x 110 -> 75 x 55 -> 38 x 28 -> 19 x 14 -> 10 x 7:
    75 -> 37.5 -> 38 and 19 -> 9.5 -> 10 round half to even upwards, 55 -> 27.5 -> 28): level sizes, the levels, the level-4 gradients"""
    f = np.ascontiguousarray(vislam.synth_frame(cv, 9, 160, 120, 123)[:110, :150])
    lw, lh = orc.half_pyramid_dims(150, 110)
    lv = orc.half_pyramid(f)
    gx4, gy4, g4 = orc.scharr_gradient(lv[4], 3)
    np.savez_compressed(os.path.join(HERE, "half_150x110.npz"), img=f, lw=np.array(lw), lh=np.array(lh),
                        levels=np.concatenate([l.ravel() for l in lv[1:]]), gx4=gx4, gy4=gy4, g4=g4)


if __name__ == "__main__":
    main()
