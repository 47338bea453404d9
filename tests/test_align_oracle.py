"""CPU: unit tests of the oracle's restatement of VISystem::EstimatePoseFeatures (oracle/align.cpp; reference
src/VISystem.cpp:1113-1448, warp :1495-1558) and of the Sophus::SE3f / cv::invert pieces it needs.  The reference holds no
fixture for this function; numpy / scipy closed forms are the hand-checkable references here."""
import numpy as np
import pytest

import align_cases


def _hat6(a):
    u, w = a[:3], a[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = u
    return M


@pytest.mark.parametrize("a", [[0, 0, 0, 0, 0, 0], [0.1, -0.2, 0.3, 0, 0, 0], [0, 0, 0, 0.02, -0.01, 0.03],
                               [0.5, 0.1, -0.4, 0.3, -0.2, 0.1], [1e-3, 2e-3, -1e-3, 1e-6, -2e-6, 3e-6], [0.2, 0.1, 0.0, 1.5, -0.7, 0.4]])
def test_se3_exp_matches_matrix_exponential(orc, a):
    from scipy.linalg import expm
    e = orc.se3_exp(a)
    M = orc.se3_matrix(e)
    ref = expm(_hat6(np.array(a, float)))
    assert np.abs(M - ref).max() < 2e-6
    q = e.as_array()[:4]
    assert abs(float((q.astype(float) ** 2).sum()) - 1) < 1e-6


def test_se3_product_and_renormalisation(orc):
    rng = np.random.default_rng(1)
    a = orc.se3_exp(rng.normal(0, 0.3, 6)); b = orc.se3_exp(rng.normal(0, 0.3, 6))
    c = orc.se3_mul(a, b)
    assert np.abs(orc.se3_matrix(c) - orc.se3_matrix(a).astype(float) @ orc.se3_matrix(b).astype(float)).max() < 1e-6
    # identity element
    i = orc.se3_exp(np.zeros(6))
    assert np.abs(orc.se3_matrix(orc.se3_mul(a, i)) - orc.se3_matrix(a)).max() < 1e-7
    # 200 chained products stay on the unit sphere (SO3::operator*= renormalises by 2 / (1 + |q|^2))
    p = a
    for _ in range(200):
        p = orc.se3_mul(p, b)
    assert abs(float((p.as_array()[:4].astype(float) ** 2).sum()) - 1) < 1e-5


def test_se3_from_rotation_matrix_round_trip(orc):
    rng = np.random.default_rng(2)
    for _ in range(20):
        e = orc.se3_exp(np.concatenate([rng.normal(0, 1, 3), rng.normal(0, 1.2, 3)]))
        M = orc.se3_matrix(e)
        f = orc.se3_from_rt(M[:3, :3], M[:3, 3])
        assert np.abs(orc.se3_matrix(f) - M).max() < 2e-6


def test_lu_invert6(orc):
    rng = np.random.default_rng(3)
    J = rng.normal(0, 1, (50, 6))
    A = (J.T @ J).astype(np.float32)
    ok, inv = orc.lu_invert6(A)
    assert ok == 1
    assert np.abs(inv.astype(float) @ A.astype(float) - np.eye(6)).max() < 1e-3
    S = A.copy(); S[3] = S[1]                     # two equal rows: singular -> cv::invert returns zeros
    ok, inv = orc.lu_invert6(S)
    assert ok == 0 and not inv.any()
    # a matrix that needs row exchanges
    P = np.eye(6, dtype=np.float32)[[1, 0, 3, 2, 5, 4]] * 2
    ok, inv = orc.lu_invert6(P)
    assert ok == 1 and np.abs(inv @ P - np.eye(6)).max() < 1e-6


def test_tukey_weights(orc):
    r = np.array([0, 1, 2, 3, 4, 200, 5, 2, 1, 0], np.float32)
    w = orc.tukey_weights(r)
    assert w[5] == 0 and 0 < w[4] < w[1] <= 1 and w[0] == w[9]


def test_identical_frames_keep_the_identity(vislam, orc, canvas):
    c = align_cases.case(vislam, orc, canvas, dx=0, dy=0, n=20)
    ap = orc.default_align_params()
    r = orc.estimate_pose_features(ap, 752, 480, c["gray1"], c["gray2"], c["gx"], c["gy"], c["cand"])
    assert list(r.iterations)[:4] == [1, 1, 1, 1]              # k = 0: zero error, zero step; k = 1: error >= last_error
    assert [float(e) for e in r.error] == [0, 0, 0, 0, 0] and r.initial_error == 0
    assert np.array_equal(r.pose.as_array(), np.array([0, 0, 0, 1, 0, 0, 0], np.float32))
    assert list(r.n_residuals)[:4] == [len(x) for x in c["cand"][:4]]


@pytest.mark.parametrize("dx,dy", [(1, 0), (2, 1), (-3, 2)])
def test_first_iteration_error_is_the_mean_squared_difference(vislam, orc, canvas, dx, dy):
    c = align_cases.case(vislam, orc, canvas, dx=dx, dy=dy, n=30)
    for lvl in range(4):
        ap = orc.default_align_params()
        ap.first_level = ap.last_level = lvl
        ap.max_iterations = 1
        r = orc.estimate_pose_features(ap, 752, 480, c["gray1"], c["gray2"], c["gx"], c["gy"], c["cand"])
        p = c["cand"][lvl].astype(int)
        direct = ((c["gray2"][lvl][p[:, 1], p[:, 0]].astype(float) - c["gray1"][lvl][p[:, 1], p[:, 0]]) ** 2).mean()
        assert r.n_residuals[lvl] == len(p) and r.iterations[lvl] == 0
        assert abs(r.error[lvl] - direct) <= 1e-6 * max(direct, 1) + 1e-3


def test_level_intrinsics_follow_initialize_pyramid(vislam, orc, canvas):
    """one candidate far from the principal point, pose = pure x translation t: the warped pixel must move by fx_l * t at
    every level (z = 1), i.e. fx halves per level (src/VISystem.cpp:1470-1471)"""
    w, h = 752, 480
    for lvl in range(4):
        cols, rows = w >> lvl, h >> lvl
        img1 = np.zeros((rows, cols), np.uint8); img2 = np.zeros((rows, cols), np.uint8)
        x0, y0 = cols // 2 + 10, rows // 2
        shift = 6
        img2[y0, x0 + shift] = 200                       # the residual is 200 exactly when the warp lands on x0 + shift
        ap = orc.default_align_params()
        ap.first_level = ap.last_level = lvl
        ap.max_iterations = 1
        fx_l = ap.fx / 2 ** lvl
        init = orc.se3_exp([shift / fx_l, 0, 0, 0, 0, 0])
        lv = [None] * 5; lv2 = [None] * 5; g = [None] * 5; cd = [None] * 5
        lv[lvl], lv2[lvl], g[lvl] = img1, img2, np.zeros((rows, cols), np.int16)
        cd[lvl] = np.array([[x0, y0, 1, 1]], np.float32)
        r = orc.estimate_pose_features(ap, w, h, lv, lv2, g, g, cd, init)
        assert r.n_residuals[lvl] == 1 and r.error[lvl] == 200.0 ** 2, lvl


def test_out_of_image_points_and_empty_lists(vislam, orc, canvas):
    c = align_cases.case(vislam, orc, canvas, dx=1, dy=1, n=10)
    ap = orc.default_align_params()
    # a pose that throws every point out of the image: zero residuals -> every level stops at k = 0, pose unchanged
    init = orc.se3_exp([50.0, 0, 0, 0, 0, 0])
    r = orc.estimate_pose_features(ap, 752, 480, c["gray1"], c["gray2"], c["gx"], c["gy"], c["cand"], init)
    assert list(r.n_residuals)[:4] == [0, 0, 0, 0] and list(r.iterations)[:4] == [0, 0, 0, 0]
    assert np.array_equal(r.pose.as_array(), init.as_array())
    empty = [np.zeros((0, 4), np.float32)] * 5
    r = orc.estimate_pose_features(ap, 752, 480, c["gray1"], c["gray2"], c["gx"], c["gy"], empty)
    assert list(r.n_residuals) == [0] * 5


def test_multi_iteration_case_and_golden(vislam, orc, canvas):
    """a case that runs several Gauss-Newton iterations per level, pinned by a committed fixture of the oracle's own output"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "align_320x240.npz"))
    ap = orc.default_align_params()
    ap.fx, ap.fy, ap.cx, ap.cy = 200.0, 200.0, 160.0, 120.0
    lv = lambda k: [g[f"{k}{l}"] for l in range(5)]
    r = orc.estimate_pose_features(ap, 320, 240, lv("gray1_"), lv("gray2_"), lv("gx_"), lv("gy_"), lv("cand_"))
    assert list(r.iterations) == list(g["iterations"]) and max(r.iterations) >= 2
    assert list(r.n_residuals) == list(g["n_residuals"])
    assert np.array_equal(np.array(r.error, np.float32), g["error"])
    assert np.array_equal(r.pose.as_array(), g["pose"])


def test_library_se3_host_helpers_equal_the_oracle(vislam, orc):
    """vis_se3_* (host-side value operations the adapters' Track() composes poses with) == the oracle, bit for bit.
    No GPU involved: these are host functions of the library."""
    rng = np.random.default_rng(7)
    for _ in range(50):
        a6 = np.concatenate([rng.normal(0, 0.5, 3), rng.normal(0, 0.8, 3) * rng.choice([1.0, 1e-6])]).astype(np.float32)
        b6 = rng.normal(0, 0.3, 6).astype(np.float32)
        ea, oa = vislam.se3_exp(a6), orc.se3_exp(a6)
        assert np.array_equal(ea.as_array(), oa.as_array())
        eb = vislam.se3_exp(b6)
        assert np.array_equal(vislam.se3_mul(ea, eb).as_array(), orc.se3_mul(oa, orc.se3_exp(b6)).as_array())
        M = vislam.se3_matrix(ea)
        assert np.array_equal(M, orc.se3_matrix(oa))
        assert np.array_equal(vislam.se3_from_rt(M[:3, :3], M[:3, 3]).as_array(), orc.se3_from_rt(M[:3, :3], M[:3, 3]).as_array())


def test_warped_point_guard_uses_the_size_of_the_indexed_mat(vislam, orc, canvas):
    """src/VISystem.cpp:1299 tests the warped point with `y2 < image2.rows && x2 < image2.cols`: image2 = grayImage[lvl] has
    Camera::Update's size (150 x 110 -> level 2: 38 x 28), one row / column more than the bookkeeping `size >> lvl` (37 x 27).  Points
    warped into column 37.x / row 27.x are residuals (the restatement rejected them with the bookkeeping size)."""
    w, h = 150, 110
    c = align_cases.case(vislam, orc, canvas, w=w, h=h, dx=1, dy=1, n=20, grad_div=8)
    pts = np.array([[36.0, y, 1.0, 1.0] for y in range(4, 26, 3)] + [[x, 26.0, 1.0, 1.0] for x in range(4, 36, 3)], np.float32)
    cand = [np.zeros((0, 4), np.float32)] * 5
    cand[2] = pts
    ap = orc.default_align_params()
    ap.fx, ap.fy, ap.cx, ap.cy = 120.0, 120.0, w / 2.0, h / 2.0
    ap.first_level, ap.last_level, ap.max_iterations = 2, 2, 1
    moved = orc.estimate_pose_features(ap, w, h, c["gray1"], c["gray2"], c["gx"], c["gy"], cand, orc.se3_exp([0.04, 0.04, 0, 0, 0, 0]))   # + 1.2 px in x and y
    assert moved.n_residuals[2] == len(pts)
    out = orc.estimate_pose_features(ap, w, h, c["gray1"], c["gray2"], c["gx"], c["gy"], cand, orc.se3_exp([0.08, 0.08, 0, 0, 0, 0]))     # + 2.4 px: beyond the Mat
    assert out.n_residuals[2] == 0
