#!/usr/bin/env python3
"""CPU-side workload statistics for the k_fast design (numpy only; no GPU): per pyramid pixel of S-752, how many
4-pixel units / pixels pass each candidate pretest, how many are FAST corners, how many survive the 3x3 NMS.
usage: tools/fast_stats.py [nframes]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import vislam  # noqa: E402
import oracle_bind as orc  # noqa: E402

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def ring_stack(img):
    h, w = img.shape
    c = img[3:h - 3, 3:w - 3].astype(np.int16)
    r = np.stack([img[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx].astype(np.int16) for dx, dy in RING])
    return c, r


def fast_score(c, r):
    """definitional: score = max(max_arc min_arc(r - c), max_arc min_arc(c - r)) - 1 over the 16 arcs of 9"""
    d = r - c[None]
    dd = np.concatenate([d, d[:8]])
    best_b = np.full(c.shape, -999, np.int16)
    best_d = np.full(c.shape, -999, np.int16)
    for k in range(16):
        a = dd[k:k + 9]
        best_b = np.maximum(best_b, a.min(0))
        best_d = np.maximum(best_d, (-a).min(0))
    return np.maximum(best_b, best_d) - 1


def main():
    nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    seed = 0xE0C00001
    cv = vislam.synth_canvas(4096, seed)
    ws, hs, sc, q = orc.level_geometry(_params(), 752, 480)
    t = 20
    tot = {}
    for fi in range(nfr):
        img = vislam.synth_frame(cv, fi * 37, 752, 480, seed)
        lv = img
        for l in range(8):
            if l > 0:
                lv = orc.resize_linear(lv, int(ws[l]), int(hs[l]))
            c, r = ring_stack(lv)
            s = fast_score(c, r)
            corner = s >= t
            sc_map = np.where(corner, s, 0)
            pad = np.pad(sc_map, 1)
            nb = np.stack([pad[1 + dy:1 + dy + sc_map.shape[0], 1 + dx:1 + dx + sc_map.shape[1]] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dx, dy) != (0, 0)])
            nms = corner & (sc_map[None] > nb).all(0)
            d = c[None] - r                                             # > t : dark ring pixel
            N, E, S, W = 0, 4, 8, 12
            dark = d > t
            bright = d < -t
            ax = ((dark[N] | dark[S]) & (dark[E] | dark[W])) | ((bright[N] | bright[S]) & (bright[E] | bright[W]))
            # 7-bit SWAR variant: c7 - r7 >= K - 1 with K = ceil(t / 2)
            c7 = c >> 1
            r7 = r >> 1
            K = (t + 1) // 2 - 1
            dark7 = (c7[None] - r7) >= K
            bright7 = (r7 - c7[None]) >= K
            ax7 = ((dark7[N] | dark7[S]) & (dark7[E] | dark7[W])) | ((bright7[N] | bright7[S]) & (bright7[E] | bright7[W]))
            # all four opposite pairs (8 ring pixels)
            def pairs(dk, br, ks):
                a = np.ones(c.shape, bool)
                b = np.ones(c.shape, bool)
                for k in ks:
                    a &= dk[k] | dk[k + 8]
                    b &= br[k] | br[k + 8]
                return a | b
            p8 = pairs(dark, bright, (0, 2, 4, 6))
            p8_7 = pairs(dark7, bright7, (0, 2, 4, 6))
            p16_7 = pairs(dark7, bright7, range(8))
            # emit region only (edge 31), as the kernel's tiles cover
            e = 31 - 3
            def reg(m):
                return m[e - 1:m.shape[0] - e + 1, e - 1:m.shape[1] - e + 1]
            def units(m):
                m = reg(m)
                wq = m.shape[1] // 4 * 4
                return m[:, :wq].reshape(m.shape[0], -1, 4).any(2).sum()
            rec = {"px": reg(ax).size, "axis": reg(ax).sum(), "axis7": reg(ax7).sum(), "pairs8": reg(p8).sum(), "pairs8_7": reg(p8_7).sum(), "pairs16_7": reg(p16_7).sum(),
                   "units_axis": units(ax), "units_axis7": units(ax7), "units_p8_7": units(p8_7), "corner": reg(corner).sum(), "units_corner": units(corner), "nms": reg(nms).sum(),
                   "level_px": lv.size}
            for k, v in rec.items():
                tot[k] = tot.get(k, 0) + int(v)
            if fi == 0:
                print(l, lv.shape, {k: int(v) for k, v in rec.items()})
    print("per frame:", {k: v / nfr for k, v in tot.items()})
    px = tot["px"]
    print("fraction of emit-region px:", {k: round(v / px, 4) for k, v in tot.items()})


def _params():
    p = orc.Params()
    for f, _ in vislam.default_params()._fields_:
        setattr(p, f, getattr(vislam.default_params(), f))
    p.nfeatures, p.nlevels, p.w_size, p.h_size = 1000, 8, 752, 480
    return p


if __name__ == "__main__":
    main()
