"""GPU parity: Hamming 2-NN + the reference's match filters vs the CPU oracle (bit-exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n1,n2", [(1000, 1000), (1, 1), (2, 1), (1, 7), (257, 63), (4000, 4000), (1000, 3), (128, 128), (129, 65), (513, 31), (255, 256)])
def test_knn2_random_descriptors(vislam, orc, ctx, n1, n2):
    rng = np.random.default_rng(n1 * 31 + n2)
    d1 = rng.integers(0, 256, (n1, 32), dtype=np.uint8)
    d2 = rng.integers(0, 256, (n2, 32), dtype=np.uint8)
    g12, g21 = ctx.bf_knn2_hamming_host(d1, d2)
    o12, o21 = orc.knn2_hamming(d1, d2)
    assert g12.tobytes() == o12.tobytes()
    assert g21.tobytes() == o21.tobytes()


def test_knn2_ties_lowest_index_first(vislam, orc, ctx):
    """many identical rows: equal distances must keep the lower train index first (cv::batchDistance)"""
    rng = np.random.default_rng(1)
    base = rng.integers(0, 256, (8, 32), dtype=np.uint8)
    d1 = base[rng.integers(0, 8, 300)]
    d2 = base[rng.integers(0, 8, 500)]
    g12, g21 = ctx.bf_knn2_hamming_host(d1, d2)
    o12, o21 = orc.knn2_hamming(d1, d2)
    assert g12.tobytes() == o12.tobytes() and g21.tobytes() == o21.tobytes()
    assert (g12["distance"][:, 0] == 0).all()


def test_knn2_extreme_distances(vislam, orc, ctx):
    """Hamming 0 and 256 (the ends of the MFMA key range: 8192 * H - 2^20 + index) next to ordinary rows"""
    rng = np.random.default_rng(5)
    d1 = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    d1[::7] = 0
    d1[3::11] = 255
    d2 = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    d2[::5] = 255
    d2[2::9] = 0
    g12, g21 = ctx.bf_knn2_hamming_host(d1, d2)
    o12, o21 = orc.knn2_hamming(d1, d2)
    assert g12.tobytes() == o12.tobytes() and g21.tobytes() == o21.tobytes()
    assert g12["distance"].min() == 0 and float(np.unpackbits(d1[3] ^ d2[2]).sum()) == 256.0


def test_knn2_empty(vislam, ctx):
    g12, g21 = ctx.bf_knn2_hamming_host(np.zeros((0, 32), np.uint8), np.zeros((5, 32), np.uint8))
    assert len(g12) == 0 and (g21["trainIdx"] == -1).all()


def test_knn2_8000_linearity(vislam, ctx):
    """full stress size (config 5): size-independent properties instead of the O(N^2) oracle:
    best distance is symmetric-consistent and matches a numpy popcount on sampled rows"""
    rng = np.random.default_rng(8)
    d1 = rng.integers(0, 256, (8000, 32), dtype=np.uint8)
    d2 = d1[rng.permutation(8000)].copy()
    flip = rng.integers(0, 256, d2.shape, dtype=np.uint8) & rng.integers(0, 256, d2.shape, dtype=np.uint8) & rng.integers(0, 256, d2.shape, dtype=np.uint8)
    d2 ^= flip & (rng.integers(0, 4, (8000, 1), dtype=np.uint8) == 0)
    g12, g21 = ctx.bf_knn2_hamming_host(d1, d2)
    pc = np.unpackbits(d1[:64, None, :] ^ d2[None, :, :], axis=2).sum(2)
    order = np.lexsort((np.arange(8000)[None, :].repeat(64, 0), pc), axis=1) if False else None
    for q in range(64):
        idx = np.lexsort((np.arange(8000), pc[q]))[:2]
        assert list(g12["trainIdx"][q]) == list(idx)
        assert list(g12["distance"][q]) == [float(pc[q][idx[0]]), float(pc[q][idx[1]])]
    assert (g12["distance"][:, 0] <= g12["distance"][:, 1]).all()


@pytest.mark.parametrize("sym_mode", [0, 1])
@pytest.mark.parametrize("t", [1, 5])
def test_good_matches_on_real_keypoints(vislam, orc, ctx, canvas, sym_mode, t):
    p = vislam.default_params()
    p.sym_mode = sym_mode
    ctx.set_params(p)
    a = vislam.synth_frame(canvas, t - 1, 752, 480)
    b = vislam.synth_frame(canvas, t, 752, 480)
    k0, d0 = ctx.orb_detect_compute(a, slot=3)
    k1, d1 = ctx.orb_detect_compute(b, slot=4)
    g12, g21 = ctx.bf_knn2_hamming(3, 4, len(k0), len(k1))
    o12, o21 = orc.knn2_hamming(d0, d1)
    assert g12.tobytes() == o12.tobytes() and g21.tobytes() == o21.tobytes()
    good, sym = ctx.good_matches(3, 4)
    og, osym = orc.good_matches(p, k0, k1, o12, o21)
    assert len(osym) > 100 and 10 < len(og) <= 49
    assert sym.tobytes() == osym.tobytes()
    assert good.tobytes() == og.tobytes()
    # host-pointer variant of the same filter chain
    good2, sym2 = ctx.good_matches_host(k0, k1, o12, o21)
    assert good2.tobytes() == og.tobytes() and sym2.tobytes() == osym.tobytes()


def test_good_matches_handmade(vislam, orc, ctx):
    """hand-made DMatch lists: ratio edge (d0 == 0.8f*d1 boundary), asymmetric best, empty result"""
    KP, DM = vislam.KEYPOINT_DTYPE, vislam.DMATCH_DTYPE
    p = vislam.default_params()
    p.w_size, p.h_size, p.n_cells = 70, 70, 49
    ctx.set_params(p)
    k1 = np.zeros(6, KP)
    k2 = np.zeros(6, KP)
    k1["x"] = [5, 15, 25, 5, 65, 69.9]
    k1["y"] = [5, 5, 5, 35, 65, 69.9]
    k2["x"] = k1["x"] + 1
    k2["y"] = k1["y"]

    def knn(best, d0, d1):
        o = np.zeros((len(best), 2), DM)
        o["queryIdx"] = np.arange(len(best))[:, None]
        o["trainIdx"][:, 0] = best
        o["trainIdx"][:, 1] = [(b + 1) % len(best) for b in best]
        o["distance"][:, 0] = d0
        o["distance"][:, 1] = d1
        return o
    k12 = knn([0, 1, 2, 3, 4, 4], [8, 80, 10, 40, 3, 3], [10, 100, 100, 50, 100, 100])
    k21 = knn([0, 1, 2, 3, 5, 5], [8, 80, 10, 40, 3, 3], [9, 81, 100, 100, 100, 4])
    for mode in (0, 1):
        p.sym_mode = mode
        ctx.set_params(p)
        g, s = ctx.good_matches_host(k1, k2, k12, k21)
        og, osym = orc.good_matches(p, k1, k2, k12, k21)
        assert g.tobytes() == og.tobytes() and s.tobytes() == osym.tobytes(), mode
    # nothing survives
    k12["distance"][:, 1] = k12["distance"][:, 0]
    g, s = ctx.good_matches_host(k1, k2, k12, k21)
    assert len(g) == 0 and len(s) == 0
