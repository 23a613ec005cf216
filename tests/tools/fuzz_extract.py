#!/usr/bin/env python3
"""Randomised parity sweep: random image sizes / pyramid depths / scale factors / thresholds / budgets AND image class
(the rectangle / disc scene or one of synth.HOSTILE_KINDS: noise, checkerboards, blob lattices, saturation, seams, ...),
GPU vs oracle, bit for bit (keypoints, descriptors, per-level counts).  usage: fuzz_extract.py [n_configs] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orbfe
import oracle_py as O
from orbfe import synth


def random_config(rng):
    W = int(rng.integers(40, 1400))
    H = int(rng.integers(40, 1100))
    scale = float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.25, 1.33, 1.5, 2.0]))
    levels = int(rng.integers(1, 13))
    while min(W, H) / scale ** (levels - 1) < 17 and levels > 1:  # every level must stay >= 16 px
        levels -= 1
    nfeat = int(rng.choice([50, 300, 1000, 1000, 2000, 5000]))
    nfast = int(rng.choice([200, 2000, 40000, 100000]))
    ini = int(rng.integers(8, 60))
    mn = int(rng.integers(1, ini + 1))
    return (nfeat, nfast, scale, levels, ini, mn, W, H)


def random_kind(rng):
    """Image class of a sweep entry: the default scene half of the time, else a hostile class."""
    return "scene" if rng.random() < 0.5 else str(rng.choice(synth.HOSTILE_KINDS))


def make_frame(kind, w, h, index):
    return synth.frame(w, h, index) if kind == "scene" else synth.hostile(kind, w, h, index)


def check(cfg, frames_per=2, seed=0, kind="scene"):
    ex = orbfe.ORBextractor(*cfg, device=0, max_batch=frames_per)
    e = O.Extractor(*cfg)
    frames = [make_frame(kind, cfg[6], cfg[7], seed * 7 + i) for i in range(frames_per)]
    got = ex.extract_batch(frames)
    one = ex.extractFeatures(frames[0])
    for i, f in enumerate(frames):
        kp_r, desc_r, per_r = e.extract(f)
        kp_g, desc_g, per_g = got[i]
        assert len(kp_g) == len(kp_r), ("count", cfg, kind, i, len(kp_g), len(kp_r))
        assert kp_g.tobytes() == kp_r.tobytes(), ("kp", cfg, kind, i)
        assert np.array_equal(desc_g, desc_r), ("desc", cfg, kind, i)
        assert np.array_equal(per_g, per_r), ("per", cfg, kind, i)
        if i == 0:
            if one is None:
                assert len(kp_r) == 0
            else:
                assert one[0].tobytes() == kp_r.tobytes() and np.array_equal(one[1], desc_r), ("single", cfg, kind)
    return sum(len(g[0]) for g in got)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    tot = 0
    for k in range(n):
        cfg = random_config(rng)
        kind = random_kind(rng)
        try:
            c = check(cfg, seed=k, kind=kind)
        except orbfe.OrbfeError as err:  # unsupported corner of the parameter space: must be the documented ones
            print("config", cfg, "->", err)
            continue
        tot += c
        print("ok", cfg, kind, c, flush=True)
    print("fuzz done:", n, "configs,", tot, "keypoints compared")
