"""GPU: the C++ adapter classes (reference CameraGPU / MatcherGPU / VISystemGPU surface over the C ABI)
driven by a main_vi_slamGPU-style frame loop produce what the oracle's per-frame pipeline produces."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_demo_loop_matches_oracle(vislam, orc, canvas):
    exe = os.path.join(ROOT, "vi-slam_amd", "lib", "vislam_demo")
    assert os.path.exists(exe), "host demo not built"
    n = 4
    out = subprocess.run([exe, str(n)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = re.findall(r"FRAME (\d+) kps (\d+) sym (\d+) good (\d+) inliers (\d+) posegood (\d+)", out.stdout)
    assert len(rows) == n
    grows = re.findall(r"GRAD (\d+) g (\d+) gx (-?\d+) patch (\d+) debug (\d+)", out.stdout)
    assert len(grows) == n
    p = vislam.default_params()
    p.fy = p.fx = float(np.float32(458.654))             # the adapters keep fx as a float member (include/VISystem.hpp)
    p.cx, p.cy = float(np.float32(367.215)), float(np.float32(248.375))
    prev = None
    for t in range(n):
        img = vislam.synth_frame(canvas, t, 752, 480)
        k, d, r = orc.pipeline_frame(p, img, prev)
        prev = (k, d)
        got = [int(x) for x in rows[t]]
        assert got[1] == len(k)
        if t > 0:
            assert got[2:] == [r.n_sym, r.n_good, r.n_inliers, r.n_pose_good], (t, got, r.n_sym, r.n_good, r.n_inliers, r.n_pose_good)
        # Camera::computeGradient of this frame: checksums over the 5 levels (weights catch transposed / shifted errors)
        gsum = gxsum = 0
        for lv in orc.half_pyramid(img):
            ox, oy, og = orc.scharr_gradient(lv)
            yy, xx = np.mgrid[0:lv.shape[0], 0:lv.shape[1]]
            gsum += int(og.astype(np.int64).sum())
            gxsum += int((ox.astype(np.int64) * (1 + ((xx + yy) & 3))).sum())
        gg = [int(x) for x in grows[t]]
        assert gg[1] == gsum and gg[2] == gxsum, (t, gg, gsum, gxsum)
        if t > 0:
            # the patch builders ran on the previous keyframe with its good matches: one debug point per match and level
            assert gg[4] == 5 * min(r.n_good, 200) and gg[3] > 0, (t, gg, r.n_good)
