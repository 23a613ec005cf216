#!/usr/bin/env python3
"""Randomised parity sweep of the two round-5 entry points against the oracle:
  orbfe_track_initialization   random geometries, image classes, initial frames (the previous frame of the scene / the same image /
                               another scene; shuffled, features dropped, only upper levels), windows, ratios, orientation flag
  orbfe_fuse_search_keyframe   random key frames, grids (8x6 ... 200x120), radii, mono / stereo, map points from the key frame's
                               own keypoints with outliers, random skip patterns travelling with the ids, ids out of range
usage: python3 tests/tools/fuzz_new_chains.py [n] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import frustum_scenarios as FS  # noqa: E402
import oracle_py as O  # noqa: E402
import orbfe  # noqa: E402
import test_fuse as TF  # noqa: E402
from orbfe import synth  # noqa: E402
from test_frustum import ON, PN  # noqa: E402


def image(kind, W, H, seed):
    if kind == "default":
        return synth.frame(W, H, seed)
    return synth.hostile(kind, W, H, seed)


def one_init(rng, k):
    W, H = int(rng.choice([320, 376, 640, 752])), int(rng.choice([240, 240, 400, 480]))
    levels = int(rng.integers(1, 9))
    nfeat = int(rng.choice([300, 1000, 1000, 2500]))
    cfg = (nfeat, 40000, 1.2, levels, 20, 7, W, H)
    kind = str(rng.choice(["default", "default", "default", "noise", "plateau", "lowtex"]))
    seed = 9000 + 10 * k
    img = image(kind, W, H, seed)
    e = O.Extractor(*cfg)
    ex = orbfe.ORBextractor(*cfg, device=0, max_batch=1)
    cols, rows = int(rng.integers(8, 100)), int(rng.integers(6, 80))
    trk = orbfe.FrameTracker(ex, cols, rows, 0.0, 0.0, float(W), float(H))
    src = str(rng.choice(["prev", "prev", "same", "other"]))
    img1 = {"prev": lambda: image(kind, W, H, seed - 1), "same": lambda: img, "other": lambda: synth.frame(W, H, seed + 5)}[src]()
    kp1, d1, _ = e.extract(img1)
    mode = str(rng.choice(["all", "all", "shuffled", "dropped", "upper", "empty"]))
    if mode == "shuffled" and len(kp1):
        p = rng.permutation(len(kp1))
        kp1, d1 = kp1[p].copy(), d1[p].copy()
    elif mode == "dropped" and len(kp1):
        keep = rng.random(len(kp1)) < 0.6
        kp1, d1 = kp1[keep].copy(), d1[keep].copy()
    elif mode == "upper":
        keep = kp1["octave"] > 0
        kp1, d1 = kp1[keep].copy(), d1[keep].copy()
    elif mode == "empty":
        kp1, d1 = kp1[:0].copy(), d1[:0].copy()
    window = int(rng.choice([10, 40, 40, 100, 300]))
    nn = float(rng.choice([0.45, 0.6, 0.9, 1.0]))
    check = bool(rng.random() < 0.7)
    ini = orbfe.InitialFrame(ex, kp1.view(orbfe.KP_DTYPE), d1)
    got = trk.TrackInitialization(img, ini, window, nn, check)
    kp, desc, per = e.extract(img)
    what = "init case %d: %s %dx%d levels %d nfeat %d grid %dx%d initial frame %s/%s (%d features) window %d nn %.2f check %d" % (
        k, kind, W, H, levels, nfeat, cols, rows, src, mode, len(kp1), window, nn, check)
    assert got["kp"].tobytes() == kp.tobytes() and np.array_equal(got["desc"], desc) and np.array_equal(got["per_level"], per), what
    if len(kp) == 0 or len(kp1) == 0:
        n_ref, m_ref = 0, np.full(len(kp1), -1, np.int32)
    else:
        fv1 = O.make_frame_view(kp1, d1, cols, rows, 0.0, 0.0, float(W), float(H), e.scaleFactors)
        fv2 = O.make_frame_view(kp, desc, cols, rows, 0.0, 0.0, float(W), float(H), e.scaleFactors)
        n_ref, m_ref = O.search_for_initialization(fv1, fv2, window, nn, check)
    assert got["nmatches"] == n_ref and np.array_equal(got["matches12"], m_ref), what + ": %d vs %d matches" % (got["nmatches"], n_ref)
    ini.close()
    ex.close()
    return len(kp), n_ref


def one_fuse(rng, k):
    W, H = 752, 480
    cfg = (int(rng.choice([300, 1000, 2000])), 40000, 1.2, 8, 20, 7, W, H)
    e = O.Extractor(*cfg)
    ex = orbfe.ORBextractor(*cfg, device=0, max_batch=1)
    m = orbfe.ORBmatcher(ex)
    kp, desc, _ = e.extract(synth.frame(W, H, 8000 + k))
    stereo = bool(rng.random() < 0.4)
    kb8 = bool(rng.random() < 0.25)
    Fo, Fp = O.Frustum(), orbfe.Frustum()
    v = FS.fill_frustum(Fo, ON, seed=300 + k, kb8=kb8)
    FS.fill_frustum(Fp, PN, seed=300 + k, kb8=kb8)
    M = int(rng.choice([1, 50, 700, 2000, 5000]))
    th = float(rng.choice([2.5, 3.0, 4.0, 8.0]))
    pts, mpd, u_right, inv_s2 = TF.scenario(kp, desc, e.scaleFactors, v, M, k, stereo)
    cols, rows = int(rng.integers(8, 200)), int(rng.integers(6, 120))
    fvo = O.make_frame_view(kp, desc, cols, rows, 0.0, 0.0, float(W), float(H), e.scaleFactors)
    bi_r, bd_r = O.fuse_search(fvo, inv_s2, u_right, Fo, th, pts, mpd)
    kf = orbfe.KeyFrame(ex, kp.view(orbfe.KP_DTYPE), desc, np.full(len(kp), -1, np.int32), e.scaleFactors)
    kf.set_grid(cols, rows, 0.0, 0.0, float(W), float(H), inv_s2, u_right)
    cap = M + int(rng.integers(0, 100))
    base = int(rng.integers(0, cap - M + 1))
    mp = orbfe.MapPoints(ex, cap)
    st = pts.copy()
    st["skip"] = 0
    mp.update(np.arange(base, base + M), st.view(orbfe.WP_DTYPE), mpd)
    ids = np.arange(base, base + M, dtype=np.int32)
    ids = np.where(pts["skip"] != 0, ~ids, ids).astype(np.int32)
    bi, bd = m.Fuse_search_keyframe(kf, mp, ids, Fp, th)
    what = "fuse case %d: N %d M %d th %.1f grid %dx%d stereo %d kb8 %d" % (k, len(kp), M, th, cols, rows, stereo, kb8)
    assert np.array_equal(bd, bd_r) and np.array_equal(bi, bi_r), what
    mp.close()
    kf.close()
    ex.close()
    return M, int((bd_r <= 30).sum())


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    tot_i = [one_init(rng, k) for k in range(n)]
    tot_f = [one_fuse(rng, k) for k in range(n)]
    print("%d initialisation chains exact (%d matches), %d resident fuse searches exact (%d fused)" % (
        n, sum(t[1] for t in tot_i), n, sum(t[1] for t in tot_f)))
