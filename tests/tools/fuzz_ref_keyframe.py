#!/usr/bin/env python3
"""Randomised parity sweep of orbfe_track_reference_keyframe (extract -> vocabulary descent -> SearchByBoW against a resident
key frame, one submission): random image sizes, pyramid depths, budgets, image classes (default scene and hostile classes),
vocabulary shapes (k, L, levelsup: from one node for the whole frame to hundreds of nodes), key frames from the same scene,
a different one or a scrambled copy, random map-point flags, ratios, orientation check on / off; every field of the result
against the oracle's chain.  usage: fuzz_ref_keyframe.py [n] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle_py as O  # noqa: E402
import orbfe  # noqa: E402
import vocab_synth as vs  # noqa: E402
from orbfe import synth  # noqa: E402

_trees = {}


def tree(k, L, seed):
    key = (k, L, seed)
    if key not in _trees:
        _trees[key] = vs.spread_first_level(vs.make_tree(k, L, seed=seed, early_leaf_p=0.04, dup_p=0.08), seed + 1)
    return _trees[key]


def csr(nodeKF, nodeF):
    kfOff, kfIdx, fOff, fIdx = [0], [], [0], []
    for g in sorted((set(nodeKF.tolist()) & set(nodeF.tolist())) - {-1}):
        kfIdx += list(np.flatnonzero(nodeKF == g))
        fIdx += list(np.flatnonzero(nodeF == g))
        kfOff.append(len(kfIdx))
        fOff.append(len(fIdx))
    return kfOff, kfIdx, fOff, fIdx


def image(kind, W, H, seed):
    return synth.frame(W, H, seed) if kind == "default" else synth.hostile(kind, W, H, seed)


def one(rng, k):
    """-> (keypoints of the frame, matches) of one random case, after asserting equality with the oracle"""
    W, H = int(rng.choice([320, 376, 640, 752])), int(rng.choice([240, 240, 400, 480]))
    levels = int(rng.integers(2, 9))
    nfeat = int(rng.choice([300, 1000, 1000, 2000]))
    cfg = (nfeat, 40000, 1.2, levels, 20, 7, W, H)
    tk, tL = [(10, 4), (10, 5), (6, 4), (4, 6), (12, 3)][int(rng.integers(0, 5))]
    levelsup = int(rng.integers(0, tL + 2))  # levelsup > L: the node level is negative, every feature in the root's node
    t = tree(tk, tL, int(rng.integers(0, 3)))
    kind = str(rng.choice(["default", "default", "default", "noise", "plateau", "checker3", "lowtex"]))
    seed = 7000 + 10 * k
    img = image(kind, W, H, seed)
    e = O.Extractor(*cfg)
    ex = orbfe.ORBextractor(*cfg, device=0, max_batch=1)
    voc = orbfe.ORBVocabulary(ex, t["childOff"], t["childIdx"], t["nodeDesc"], t["wordId"], t["weight"], tL)
    trk = orbfe.FrameTracker(ex, 16, 12, 0.0, 0.0, float(W), float(H))
    tv = lambda d: O.vocab_transform(t["childOff"], t["childIdx"], t["nodeDesc"], t["wordId"], t["weight"], tL, d, levelsup)
    # the key frame: the next frame of the same scene / the same image / another class, possibly with a few bits flipped and shuffled
    src = str(rng.choice(["next", "next", "same", "other"]))
    kimg = {"next": lambda: image(kind, W, H, seed + 1), "same": lambda: img, "other": lambda: synth.frame(W, H, seed + 5)}[src]()
    kkp, kdesc, _ = e.extract(kimg)
    if len(kkp) and rng.random() < 0.5:
        perm = rng.permutation(len(kkp))
        kkp, kdesc = kkp[perm].copy(), kdesc[perm].copy()
        flip = rng.random(len(kkp)) < 0.3
        kdesc[flip, int(rng.integers(0, 32))] ^= np.uint8(1 << int(rng.integers(0, 8)))
    if len(kkp):  # the key frame's mFeatVec: features on stopped words (weight 0) are in no node (TemplatedVocabulary.h:1168-1172)
        _, kn_, kw_ = tv(kdesc)
        knode = np.where(kw_ > 0, kn_, -1).astype(np.int32)
    else:
        knode = np.zeros(0, np.int32)
    if len(knode) and rng.random() < 0.3:
        knode[rng.random(len(knode)) < 0.05] = -1  # features the key frame's FeatureVector does not hold
    has = (rng.random(len(kkp)) < float(rng.choice([0.3, 0.7, 1.0]))).astype(np.uint8)
    nn = float(rng.choice([0.6, 0.75, 0.9, 1.0]))
    check = bool(rng.random() < 0.7)
    res = orbfe.KeyFrame(ex, kkp.view(orbfe.KP_DTYPE), kdesc, knode, e.scaleFactors)
    got = trk.TrackReferenceKeyFrame(img, voc, levelsup, res, has, nn, check)
    kp, desc, per = e.extract(img)
    word, node, weight = tv(desc)
    what = "case %d: %s %dx%d levels %d nfeat %d tree (%d, %d) levelsup %d key frame %s (%d features) nn %.2f check %d" % (
        k, kind, W, H, levels, nfeat, tk, tL, levelsup, src, len(kkp), nn, check)
    assert got["kp"].tobytes() == kp.tobytes() and np.array_equal(got["desc"], desc) and np.array_equal(got["per_level"], per), what
    assert np.array_equal(got["word"], word) and np.array_equal(got["node"], node), what
    assert np.array_equal(got["weight"].view(np.uint64), weight.view(np.uint64)), what
    if len(kp) == 0 or len(kkp) == 0:
        n_ref, m_ref = 0, np.full(len(kp), -1, np.int32)
    else:
        n_ref, m_ref = O.search_by_bow(*csr(knode, np.where(weight > 0, node, -1)), kdesc, kkp["angle"], has, desc, kp["angle"], nn, check)
    assert got["nmatches"] == n_ref and np.array_equal(got["match"], m_ref), what + ": %d vs %d matches" % (got["nmatches"], n_ref)
    voc.close()
    res.close()
    ex.close()
    return len(kp), n_ref


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    tot = [one(rng, k) for k in range(n)]
    print("fuzz_ref_keyframe: %d cases exact, %d keypoints, %d matches" % (n, sum(a for a, _ in tot), sum(b for _, b in tot)))
