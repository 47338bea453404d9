"""GPU: the larger BASELINE.json configurations.
 config 3: 1920x1080, 4 levels, 4000 kps + essential RANSAC with a fixed 2000 iterations
 config 5: 3840x2160, 8000 kps, 8000x8000 all-pairs
Parity against the oracle where it finishes in seconds, size-independent properties otherwise."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big_canvas(vislam):
    return vislam.synth_canvas(4096, 0xE0C00003)


def test_config3_1080p_4levels_4000kps(vislam, orc, big_canvas):
    p = vislam.default_params()
    p.nfeatures, p.nlevels, p.w_size, p.h_size = 4000, 4, 1920, 1080
    p.ransac_adaptive, p.ransac_max_iters = 0, 2000
    p.fy = p.fx
    c = vislam.Context(0, p)
    a = vislam.synth_frame(big_canvas, 0, 1920, 1080, 0xE0C00003)
    b = vislam.synth_frame(big_canvas, 1, 1920, 1080, 0xE0C00003)
    k0, d0 = c.orb_detect_compute(a, slot=0, cap=8192)
    k1, d1 = c.orb_detect_compute(b, slot=1, cap=8192)
    ok0, od0 = orc.orb_detect_compute(p, a, cap=8192)
    assert len(k0) >= 3900 and k0.tobytes() == ok0.tobytes() and (d0 == od0).all()
    g12, g21 = c.bf_knn2_hamming(0, 1, len(k0), len(k1))
    o12, o21 = orc.knn2_hamming(d0, d1)
    assert g12.tobytes() == o12.tobytes() and g21.tobytes() == o21.tobytes()
    good, sym = c.good_matches(0, 1)
    og, osym = orc.good_matches(p, k0, k1, o12, o21)
    assert good.tobytes() == og.tobytes() and sym.tobytes() == osym.tobytes() and len(sym) > 1500
    # RANSAC on the un-gridded symmetric matches (M ~ thousands), 2000 fixed iterations
    p1 = np.stack([k0["x"][sym["queryIdx"]], k0["y"][sym["queryIdx"]]], 1)
    p2 = np.stack([k1["x"][sym["trainIdx"]], k1["y"][sym["trainIdx"]]], 1)
    E, mask, ninl, iters = c.essential_ransac(p1, p2)
    oE, omask, oninl, oiters = orc.essential_ransac(p, p1, p2)
    assert iters == oiters == 2000 and ninl == oninl and (mask == omask).all()
    s = 1.0 if float((E * oE).sum()) >= 0 else -1.0
    assert np.abs(E - s * oE).max() <= 1e-9
    c.close()


def test_config5_2160p_8000kps_properties(vislam, orc, big_canvas):
    p = vislam.default_params()
    p.nfeatures, p.nlevels, p.w_size, p.h_size = 8000, 8, 3840, 2160
    c = vislam.Context(0, p)
    a = vislam.synth_frame(big_canvas, 0, 3840, 2160, 0xE0C00003)
    k0, d0 = c.orb_detect_compute(a, slot=0, cap=16384)
    k1, d1 = c.orb_detect_compute(a, slot=1, cap=16384)          # same image twice: idempotence
    assert len(k0) >= 7900 and k0.tobytes() == k1.tobytes() and (d0 == d1).all()
    ws, hs, sc, q = c.level_geometry(3840, 2160)
    # canonical order: octave ascending, response descending inside an octave; per-level counts >= quota unless starved
    assert (np.diff(k0["octave"]) >= 0).all()
    for l in range(8):
        r = k0["response"][k0["octave"] == l]
        assert (np.diff(r) <= 0).all() and len(r) >= q[l]
    # keypoints respect the 31 px border in their own level
    x_l = k0["x"] / sc[k0["octave"]]
    y_l = k0["y"] / sc[k0["octave"]]
    assert (x_l >= 30.5).all() and (x_l < ws[k0["octave"]] - 30.5).all() and (y_l >= 30.5).all() and (y_l < hs[k0["octave"]] - 30.5).all()
    # 8000 x 8000 all-pairs on identical sets: every row's best match is itself at distance 0
    g12, g21 = c.bf_knn2_hamming(0, 1, len(k0), len(k1))
    assert (g12["distance"][:, 0] == 0).all() and (g12["distance"][:, 0] <= g12["distance"][:, 1]).all()
    dup = g12["trainIdx"][:, 0] != np.arange(len(k0))
    assert (g12["distance"][dup, 1] == 0).all()                 # only exact duplicates may rank a lower index first
    assert g12.tobytes() == g21.tobytes()
    # oracle parity on the full frame is still affordable once
    ok0, od0 = orc.orb_detect_compute(p, a, cap=16384)
    assert k0.tobytes() == ok0.tobytes() and (d0 == od0).all()
    good, sym = c.good_matches(0, 1)
    assert len(sym) > 7000 and len(good) == 49
    c.close()
