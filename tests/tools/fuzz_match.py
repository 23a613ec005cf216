#!/usr/bin/env python3
"""Randomised parity sweep of SearchByProjection: random frame sizes, grids, radii, ratios, map-point counts,
initial claims and far-point filters; GPU vs oracle, exact match indices.  usage: fuzz_match.py [n] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orbfe
import oracle_py as O
import match_scenarios as S
from orbfe import synth

NAMES_O = ("projX", "projY", "viewCos", "trackDepth", "level", "inView", "bad", "observations")


def one(rng, k):
    W, H = int(rng.integers(160, 1300)), int(rng.integers(120, 800))
    if round(W / H) == 0:
        W = H
    levels = int(rng.integers(1, 9))
    nfeat = int(rng.choice([200, 1000, 1000, 3000]))
    cfg = (nfeat, 40000, 1.2, levels, 20, 7, W, H)
    e = O.Extractor(*cfg)
    kp, desc, _ = e.extract(synth.frame(W, H, 500 + k))
    if len(kp) == 0:
        return None
    ex = orbfe.ORBextractor(*cfg)
    cols, rows = int(rng.choice([8, 16, 64, 64, 64, 100, 257])), int(rng.choice([6, 12, 48, 48, 48, 75, 130]))
    th = float(rng.choice([1.0, 3.0, 6.0, 20.0, 40.0]))
    nn = float(rng.choice([0.6, 0.75, 0.85, 0.9, 1.0]))
    M = int(rng.choice([1, 50, 700, 2000, 4000]))
    mps, mpd, init_obs = S.projection_scenario(kp, desc, M, int(rng.integers(1 << 30)), O.MP_DTYPE, NAMES_O, e.nLevels)
    if rng.random() < 0.4:
        init_obs = None
    far = bool(rng.random() < 0.3)
    thfar = float(rng.uniform(2, 20))
    fvo = O.make_frame_view(kp, desc, cols, rows, 0.0, 0.0, float(W), float(H), e.scaleFactors)
    n_ref, out_ref = O.search_by_projection(fvo, mps, mpd, init_obs, th, nn, far, thfar)
    fv = orbfe.make_frame_view(kp, desc, cols, rows, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
    n, out = orbfe.ORBmatcher(ex).SearchByProjection(fv, mps.view(orbfe.MP_DTYPE), mpd, th, far, thfar, nn, init_obs)
    assert n == n_ref and np.array_equal(out, out_ref), ("mismatch", cfg, cols, rows, th, nn, M, far)
    return (W, H, levels, len(kp), cols, rows, th, nn, M, far, n_ref)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    for k in range(n):
        print("ok", one(rng, k), flush=True)
    print("fuzz_match done:", n)
