#!/usr/bin/env python3
"""How much of FAST stage A's queue would an 8-point (even ring positions) pre-test reject?  CPU study on the oracle's
pyramid levels (profiles/r04_fast_prefilter_study.json).  A 9-arc of the 16-ring contains >= 4 consecutive even positions,
so "4 circularly consecutive even ring points all brighter than v + th (or all darker than v - th)" is an exact reject."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_py as O  # noqa: E402
from orbfe import synth  # noqa: E402

ARGS = (1000, 40000, 1.2, 8, 20, 7, 752, 480)
W, H = 752, 480
RING = [(3, 0), (3, 1), (2, 2), (1, 3), (0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0), (-3, -1), (-2, -2), (-1, -3), (0, -3), (1, -3), (2, -2), (3, -1)]


def stats(img, th):
    h, w = img.shape
    im = img.astype(np.int16)
    c = im[6:h - 5, 6:w - 5]
    ring = [im[6 + di:h - 5 + di, 6 + dj:w - 5 + dj] for di, dj in RING]
    s_, e_, n_, w_ = ring[0], ring[4], ring[8], ring[12]
    queue_a = (np.minimum(np.maximum(n_, s_), np.maximum(e_, w_)) - c > th) | (c - np.maximum(np.minimum(n_, s_), np.minimum(e_, w_)) > th)

    def run(masks, length):
        out = np.zeros_like(masks[0])
        for s in range(len(masks)):
            a = masks[s].copy()
            for t in range(1, length):
                a &= masks[(s + t) % len(masks)]
            out |= a
        return out

    ev = [ring[k] for k in range(0, 16, 2)]
    pass8 = run([e - c > th for e in ev], 4) | run([c - e > th for e in ev], 4)
    corners = run([r - c > th for r in ring], 9) | run([c - r > th for r in ring], 9)
    assert not (corners & ~pass8).any() and not (corners & ~queue_a).any()
    return np.array([queue_a.sum(), (queue_a & pass8).sum(), corners.sum(), queue_a.size], np.float64)


if __name__ == "__main__":
    eo = O.Extractor(*ARGS)
    for name, gen in (("default", lambda: synth.stream(W, H, 3, index0=5)), ("pink", lambda: synth.pink_stream(W, H, 2, index0=5)),
                      ("lowtex", lambda: synth.lowtex_stream(W, H, 2, index0=5))):
        tot = np.zeros(4)
        for img in gen():
            eo.extract(img)
            for lvl in range(8):
                tot += stats(eo.level_image(lvl, False), 7)
        a, p, c, n = tot
        print("%-8s positions %d  queue A %.1f %% of positions; pass8 %.1f %% of A (rejected %.1f %%); corners %.1f %% of A" % (
            name, n, 100 * a / n, 100 * p / a, 100 * (1 - p / a), 100 * c / a))
