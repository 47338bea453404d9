"""CPU: vi::CameraModel with a rectifying calibration (src/CameraModel.cpp:84-88, src/VISystemGPU.cpp:60-76, src/VISystem.cpp:162-205).
The adapter restates cv::getOptimalNewCameraMatrix(alpha = 1) and derives CalculateROI's rectangle from the rectification map; an
independent numpy / scipy evaluation (exact inverse of the distortion model by root finding instead of the 5 fixed-point
iterations) has to agree.  OpenCV itself is absent: this pins the restatement against the published model, not against calib3d
(PARITY UNPINNED)."""
import json
import os
import subprocess

import numpy as np
import pytest
from scipy.optimize import fsolve

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "vi-slam_amd", "lib", "camera_model_probe")

XML = """<?xml version="1.0"?>
<opencv_storage>
<in_width type_id="integer"> {iw} </in_width>
<in_height type_id="integer"> {ih} </in_height>
<out_width type_id="integer"> {ow} </out_width>
<out_height type_id="integer"> {oh} </out_height>
<calibration_values type_id="opencv-matrix"><rows>1</rows><cols>4</cols><dt>f</dt>
  <data> {fx} {fy} {cx} {cy} </data></calibration_values>
<rectification type_id="opencv-matrix"><rows>1</rows><cols>4</cols><dt>f</dt>
  <data> {k1} {k2} {p1} {p2} </data></rectification>
<num_cells type_id="integer"> 49</num_cells>
</opencv_storage>
"""

CASES = [
    dict(iw=752, ih=480, ow=736, oh=480, fx=458.654, fy=457.296, cx=367.215, cy=248.375, k1=-0.28340811, k2=0.07395907, p1=0.00019359, p2=1.76187114e-05),   # calibrationEUROC.xml
    dict(iw=640, ih=480, ow=640, oh=480, fx=520.0, fy=515.0, cx=318.0, cy=243.0, k1=-0.12, k2=0.03, p1=-0.001, p2=0.0007),
    dict(iw=1280, ih=720, ow=1024, oh=576, fx=900.0, fy=905.0, cx=650.0, cy=350.0, k1=0.05, k2=-0.01, p1=0.0, p2=0.0),                                       # pincushion
]


def _distort(x, y, k1, k2, p1, p2):
    r2 = x * x + y * y
    kr = 1 + k1 * r2 + k2 * r2 * r2
    return x * kr + 2 * p1 * x * y + p2 * (r2 + 2 * x * x), y * kr + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y


def _probe(tmp_path, c):
    f = tmp_path / "cal.xml"
    f.write_text(XML.format(**c))
    out = subprocess.run([EXE, str(f)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("c", CASES)
def test_optimal_new_camera_matrix_and_roi(tmp_path, built, c):
    j = _probe(tmp_path, c)
    assert j["valid"] == 1 and j["in"] == [c["iw"], c["ih"]] and j["out"] == [c["ow"], c["oh"]]
    k = [np.float32(c[n]) for n in ("k1", "k2", "p1", "p2")]
    fx, fy, cx, cy = (np.float64(np.float32(c[n])) for n in ("fx", "fy", "cx", "cy"))
    # exact inverse of the distortion model on the 9 x 9 grid (the adapter, like calib3d, runs 5 fixed-point iterations)
    pts = []
    for gy in range(9):
        for gx in range(9):
            u, v = np.float32(gx) * c["iw"] / 8, np.float32(gy) * c["ih"] / 8
            xd, yd = (u - cx) / fx, (v - cy) / fy
            sol = fsolve(lambda q: np.subtract(_distort(q[0], q[1], *map(float, k)), (xd, yd)), (xd, yd), xtol=1e-12)
            pts.append(sol)
    pts = np.array(pts)
    ox0, oy0 = pts.min(0); ox1, oy1 = pts.max(0)
    fxn, fyn = (c["ow"] - 1) / (ox1 - ox0), (c["oh"] - 1) / (oy1 - oy0)
    ref = np.array([fxn, fyn, -fxn * ox0, -fyn * oy0])
    got = np.array(j["K"])
    # 5 iterations of the fixed point leave ~1e-3 of relative error at the image corners of a strongly distorted lens
    assert np.allclose(got, ref, rtol=5e-3, atol=0.5), (got, ref)
    # the ROI from the rectification map, re-evaluated here with the probe's own K'
    fxo, fyo, cxo, cyo = got

    def inside(u, v):
        sx, sy = _distort((u - cxo) / fxo, (v - cyo) / fyo, *map(float, k))
        su, sv = fx * sx + cx, fy * sy + cy
        return -1.0 < su < c["iw"] and -1.0 < sv < c["ih"]
    xm, ym = int((c["ow"] - 1) * 0.5), int((c["oh"] - 1) * 0.5)
    x1 = next(x for x in range(c["ow"]) if inside(x, ym)); x2 = next(x for x in range(c["ow"] - 1, -1, -1) if inside(x, ym))
    y1 = next(y for y in range(c["oh"]) if inside(xm, y)); y2 = next(y for y in range(c["oh"] - 1, -1, -1) if inside(xm, y))
    assert j["roi"] == [x1 + 5, y1 + 5, x2 - 5, y2 - 5]
    assert 0 < j["roi"][0] < j["roi"][2] < c["ow"] and 0 < j["roi"][1] < j["roi"][3] < c["oh"]


def test_no_distortion_keeps_the_original_intrinsics(tmp_path, built):
    c = dict(CASES[0], k1=0.0, k2=0.0, p1=0.0, p2=0.0)
    j = _probe(tmp_path, c)
    assert j["valid"] == 0 and j["K"] == j["K0"]                    # src/CameraModel.cpp:78-83
