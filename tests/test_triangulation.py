"""SURVEY.md section 8f row f2 (second half): ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:441-676)."""
import numpy as np
import pytest

import match_scenarios as S
import oracle_py as O

W, H = 752, 480
ARGS = (1000, 40000, 1.2, 8, 20, 7, W, H)
f32 = np.float32
TH_LOW = 30  # include/ORBmatcher.h:73 (this fork; upstream ORB-SLAM3 uses 50)


def scenario(kp, desc, seed, consistent, stereo):
    """Key frame 2 = key frame 1 seen from a camera translated along x: same rows (+ noise), shifted columns,
    descriptors with a few flipped bits, a shuffled order, plus unrelated features.  Vocabulary nodes follow the
    position so that corresponding features share a node (with some strays)."""
    rng = np.random.default_rng(seed)
    n1 = len(kp)
    keep = rng.random(n1) < 0.8
    src = np.flatnonzero(keep)
    kp2 = kp[src].copy()
    kp2["x"] = kp2["x"] - rng.uniform(2, 40, len(src)).astype(np.float32)
    kp2["y"] = kp2["y"] + rng.normal(0, 1.3, len(src)).astype(np.float32)
    kp2["angle"] = (kp2["angle"] + rng.choice([0, 0, 0, 90], len(src)) + rng.normal(0, 2, len(src))).astype(np.float32) % 360
    d2 = np.stack([S.flip_bits(desc[s], int(rng.integers(0, 36)), rng) for s in src])
    extra = rng.integers(0, n1, n1 // 5)
    kpe = kp[extra].copy()
    kpe["x"] = rng.uniform(10, W - 10, len(extra))
    kp2 = np.concatenate([kp2, kpe])
    d2 = np.concatenate([d2, rng.integers(0, 256, (len(extra), 32), dtype=np.uint8)])
    perm = rng.permutation(len(kp2))
    kp2, d2 = kp2[perm], d2[perm]
    node1 = (kp["y"] // 40).astype(int) * 8 + kp["octave"]
    node2 = np.concatenate([node1[src], rng.integers(0, node1.max() + 1, len(extra))])[perm]
    node2 = np.where(rng.random(len(node2)) < 0.05, rng.integers(0, node1.max() + 1, len(node2)), node2)
    common = sorted(set(node1) & set(node2))
    off1, idx1, off2, idx2 = [0], [], [0], []
    for g in common:
        i1 = np.flatnonzero(node1 == g)
        i2 = np.flatnonzero(node2 == g)
        idx1 += list(i1)
        idx2 += list(rng.permutation(i2))  # FeatureVector order is insertion order == ascending index; any order must work
        off1.append(len(idx1))
        off2.append(len(idx2))
    has1 = rng.random(n1) < 0.3
    has2 = rng.random(len(kp2)) < 0.3
    s1 = (rng.random(n1) < 0.4) if stereo else None
    s2 = (rng.random(len(kp2)) < 0.4) if stereo else None
    if consistent:
        K = np.array([[458.654, 0, 367.215], [0, 457.296, 248.375], [0, 0, 1]])
        tx = np.array([[0, 0, 0], [0, 0, -0.11], [0, 0.11, 0]])
        F12 = (np.linalg.inv(K).T @ tx @ np.linalg.inv(K)).astype(np.float32)
        ep = (-5000.0, 248.0)
    else:
        F12 = rng.normal(0, 1e-3, (3, 3)).astype(np.float32)
        F12[2, 2] = 0.3
        ep = (400.0, 240.0)
    return off1, idx1, off2, idx2, kp2, d2, has1, has2, s1, s2, F12, ep


def py_tri(off1, idx1, off2, idx2, kp1, d1, h1, s1, kp2, d2, h2, s2, sf2, F12, ep, only_stereo, coarse, check):
    """Plain restatement of :441-676 (pins the C oracle)."""
    F = np.asarray(F12, f32)
    out = np.full(len(kp1), -1, np.int64)
    bins = {}
    for g in range(len(off1) - 1):
        for a1 in idx1[off1[g]:off1[g + 1]]:
            if h1[a1]:
                continue
            st1 = bool(s1 is not None and s1[a1])
            if only_stereo and not st1:
                continue
            k1 = kp1[a1]
            best, best2 = TH_LOW, -1
            for a2 in idx2[off2[g]:off2[g + 1]]:
                if h2[a2]:
                    continue
                st2 = bool(s2 is not None and s2[a2])
                if only_stereo and not st2:
                    continue
                dist = int(np.unpackbits(d1[a1] ^ d2[a2]).sum())
                if dist > TH_LOW or dist > best:
                    continue
                k2 = kp2[a2]
                if not st1 and not st2:
                    ex, ey = f32(f32(ep[0]) - k2["x"]), f32(f32(ep[1]) - k2["y"])
                    if f32(f32(ex * ex) + f32(ey * ey)) < f32(f32(100) * sf2[k2["octave"]]):
                        continue
                a = f32(f32(f32(k1["x"] * F[0, 0]) + f32(k1["y"] * F[1, 0])) + F[2, 0])
                b = f32(f32(f32(k1["x"] * F[0, 1]) + f32(k1["y"] * F[1, 1])) + F[2, 1])
                c = f32(f32(f32(k1["x"] * F[0, 2]) + f32(k1["y"] * F[1, 2])) + F[2, 2])
                num = f32(f32(f32(a * k2["x"]) + f32(b * k2["y"])) + c)
                den = f32(f32(a * a) + f32(b * b))
                ok = den != 0 and float(f32(f32(num * num) / den)) < 3.84
                if coarse or ok:
                    best, best2 = dist, a2
            if best2 >= 0:
                out[a1] = best2
                if check:
                    rot = f32(k1["angle"] - kp2[best2]["angle"])
                    if rot < 0:
                        rot = f32(rot + f32(360))
                    b_ = int(np.floor(float(f32(rot * f32(1.0 / 30))) + 0.5))  # roundf for non-negative values
                    bins.setdefault(0 if b_ == 30 else b_, []).append(a1)
    if check:
        sizes = [len(bins.get(i, [])) for i in range(30)]
        m = [0, 0, 0]
        ind = [-1, -1, -1]
        for i, s_ in enumerate(sizes):
            if s_ > m[0]:
                m, ind = [s_, m[0], m[1]], [i, ind[0], ind[1]]
            elif s_ > m[1]:
                m, ind = [m[0], s_, m[1]], [ind[0], i, ind[1]]
            elif s_ > m[2]:
                m[2], ind[2] = s_, i
        if f32(m[1]) < f32(0.1) * f32(m[0]):
            ind[1] = ind[2] = -1
        elif f32(m[2]) < f32(0.1) * f32(m[0]):
            ind[2] = -1
        for i in range(30):
            if i not in ind:
                for a1 in bins.get(i, []):
                    out[a1] = -1
    return int((out >= 0).sum()), out


CASES = [(1, True, False, False, False, True), (2, False, True, False, False, True), (3, True, True, True, False, False),
         (4, False, False, False, True, True)]


@pytest.mark.parametrize("seed,consistent,stereo,only_stereo,coarse,check", CASES)
def test_oracle_triangulation_matches_restatement(built, seed, consistent, stereo, only_stereo, coarse, check):
    from orbfe import synth
    eo = O.Extractor(*ARGS)
    kp, desc, _ = eo.extract(synth.frame(W, H, 30 + seed))
    kp, desc = kp[:400], desc[:400]
    off1, idx1, off2, idx2, kp2, d2, h1, h2, s1, s2, F12, ep = scenario(kp, desc, seed, consistent, stereo)
    n, out = O.search_for_triangulation(off1, idx1, off2, idx2, kp, desc, h1, s1, kp2, d2, h2, s2, eo.scaleFactors, F12, ep,
                                        only_stereo, coarse, check)
    n_py, out_py = py_tri(off1, idx1, off2, idx2, kp, desc, h1, s1, kp2, d2, h2, s2, eo.scaleFactors, F12, ep, only_stereo,
                          coarse, check)
    assert n == n_py and np.array_equal(out, out_py)
    if consistent or coarse:
        assert n > 15


@pytest.mark.gpu
@pytest.mark.parametrize("seed,consistent,stereo,only_stereo,coarse,check", CASES + [(5, True, False, False, False, False)])
def test_gpu_triangulation_matches_oracle(built, seed, consistent, stereo, only_stereo, coarse, check):
    import orbfe
    from orbfe import synth
    eo = O.Extractor(*ARGS)
    kp, desc, _ = eo.extract(synth.frame(W, H, 30 + seed))
    ex = orbfe.ORBextractor(*ARGS)
    m = orbfe.ORBmatcher(ex)
    off1, idx1, off2, idx2, kp2, d2, h1, h2, s1, s2, F12, ep = scenario(kp, desc, seed, consistent, stereo)
    n_ref, out_ref = O.search_for_triangulation(off1, idx1, off2, idx2, kp, desc, h1, s1, kp2, d2, h2, s2, eo.scaleFactors,
                                                F12, ep, only_stereo, coarse, check)
    n, out = m.SearchForTriangulation(off1, idx1, off2, idx2, kp.view(orbfe.KP_DTYPE), desc, h1, s1, kp2.view(orbfe.KP_DTYPE),
                                      d2, h2, s2, ex.mvScaleFactor, F12, ep, only_stereo, coarse, check)
    assert n == n_ref and np.array_equal(out, out_ref)
    if consistent:
        assert n_ref > 30
    # no common node at all
    n0, out0 = m.SearchForTriangulation([0], [], [0], [], kp.view(orbfe.KP_DTYPE), desc, h1, s1, kp2.view(orbfe.KP_DTYPE), d2,
                                        h2, s2, ex.mvScaleFactor, F12, ep)
    assert n0 == 0 and (out0 == -1).all()
