"""The mapping thread's call shape of SearchForTriangulation: ONE key frame against K neighbours
(LocalMapping::CreateNewMapPoints, src/LocalMapping.cc:455-488), through resident key frames + one launch
(orbfe_match_triangulation_batch) + the host replay (orbfe_triangulation_select).

Reference semantics = K SEQUENTIAL calls, between which the matches that survive the host's triangulation checks
(:500-700) become map points of key frame 1 -- features that received one are skipped by the later calls
(src/ORBmatcher.cc:506-509).  The tests run that sequence on the oracle (one O.search_for_triangulation per neighbour with
the flags as they stand) and compare every neighbour's (nmatches, vMatches12) with the batch + replay path."""
import numpy as np
import pytest

import match_scenarios as S
import oracle_py as O

W, H = 752, 480
ARGS = (1000, 40000, 1.2, 8, 20, 7, W, H)


def nodes_of(kp):
    return ((kp["y"] // 40).astype(np.int32) * 8 + kp["octave"]).astype(np.int32)


def neighbour(kp, desc, seed, consistent, stereo):
    """a neighbour key frame of (kp, desc): most features re-observed (shifted, noisy, a few bits flipped), some strangers,
    shuffled; vocabulary nodes follow the position, 5 % stray, 3 % of the features are in no node at all"""
    rng = np.random.default_rng(seed)
    n1 = len(kp)
    src = np.flatnonzero(rng.random(n1) < 0.8)
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
    node1 = nodes_of(kp)
    node2 = np.concatenate([node1[src], rng.integers(0, node1.max() + 1, len(extra))])[perm].astype(np.int32)
    node2 = np.where(rng.random(len(node2)) < 0.05, rng.integers(0, node1.max() + 1, len(node2)), node2).astype(np.int32)
    node2[rng.random(len(node2)) < 0.03] = -1
    has2 = (rng.random(len(kp2)) < 0.3).astype(np.uint8)
    s2 = (rng.random(len(kp2)) < 0.4).astype(np.uint8) if stereo else None
    if consistent:
        Kc = np.array([[458.654, 0, 367.215], [0, 457.296, 248.375], [0, 0, 1]])
        tx = np.array([[0, 0, 0], [0, 0, -0.11], [0, 0.11, 0]])
        F12 = (np.linalg.inv(Kc).T @ tx @ np.linalg.inv(Kc)).astype(np.float32)
        ep = (-5000.0, 248.0)
    else:
        F12 = rng.normal(0, 1e-3, (3, 3)).astype(np.float32)
        F12[2, 2] = 0.3
        ep = (400.0, 240.0)
    return dict(kp=kp2, desc=d2, node=node2, has=has2, stereo=s2, F12=F12, ep=ep)


def csr(node1, node2):
    """the merge-walk of the two FeatureVectors (src/ORBmatcher.cc:489-617): shared nodes ascending, features ascending"""
    off1, idx1, off2, idx2 = [0], [], [0], []
    for g in sorted((set(node1) & set(node2)) - {-1}):
        idx1 += list(np.flatnonzero(node1 == g))
        idx2 += list(np.flatnonzero(node2 == g))
        off1.append(len(idx1))
        off2.append(len(idx2))
    return off1, idx1, off2, idx2


def sequential_reference(kp, desc, node1, has1, s1, nbs, sf, only_stereo, coarse, check, keep_seed):
    """K oracle calls in neighbour order; after each, a pseudo-random 70 % of its matches "triangulate" and become map
    points of key frame 1 (the host's checks of src/LocalMapping.cc:500-700 stand in as a coin per match)"""
    rng = np.random.default_rng(keep_seed)
    has = np.array(has1, np.uint8).copy()
    out = []
    for nb in nbs:
        off1, idx1, off2, idx2 = csr(node1, nb["node"])
        n, m12 = O.search_for_triangulation(off1, idx1, off2, idx2, kp, desc, has, s1, nb["kp"], nb["desc"], nb["has"], nb["stereo"],
                                            sf, nb["F12"], nb["ep"], only_stereo, coarse, check)
        out.append((n, m12.copy()))
        new = np.flatnonzero(m12 >= 0)
        new = new[rng.random(len(new)) < 0.7]
        has[new] = 1
    return out


def _frame(seed):
    from orbfe import synth
    eo = O.Extractor(*ARGS)
    kp, desc, _ = eo.extract(synth.frame(W, H, 30 + seed))
    return eo, kp, desc


@pytest.mark.gpu
@pytest.mark.parametrize("K,stereo,only_stereo,coarse,check", [(20, False, False, False, True), (7, True, False, False, True),
                                                               (5, True, True, False, False), (3, False, False, True, True),
                                                               (1, False, False, False, True)])
def test_batch_plus_replay_equals_k_sequential_calls(built, K, stereo, only_stereo, coarse, check):
    import orbfe
    eo, kp, desc = _frame(K)
    ex = orbfe.ORBextractor(*ARGS)
    rng = np.random.default_rng(100 + K)
    node1 = nodes_of(kp)
    node1[rng.random(len(kp)) < 0.03] = -1
    has1 = (rng.random(len(kp)) < 0.3).astype(np.uint8)
    s1 = (rng.random(len(kp)) < 0.4).astype(np.uint8) if stereo else None
    nbs = [neighbour(kp, desc, 1000 * K + k, consistent=(k % 4 != 3), stereo=stereo) for k in range(K)]
    ref = sequential_reference(kp, desc, node1, has1, s1, nbs, eo.scaleFactors, only_stereo, coarse, check, 5)
    kf1 = orbfe.KeyFrame(ex, kp.view(orbfe.KP_DTYPE), desc, node1, ex.mvScaleFactor, s1)
    kf2 = [orbfe.KeyFrame(ex, nb["kp"].view(orbfe.KP_DTYPE), nb["desc"], nb["node"], ex.mvScaleFactor, nb["stereo"]) for nb in nbs]
    params = [orbfe.tri_params(nb["F12"], nb["ep"], only_stereo, coarse, check) for nb in nbs]
    raw, rbin = orbfe.SearchForTriangulation_batch(ex, kf1, has1, kf2, [nb["has"] for nb in nbs], params)
    rng2 = np.random.default_rng(5)  # the same coins as sequential_reference
    has = has1.copy()
    total = 0
    for k in range(K):
        n, m12 = orbfe.triangulation_select(raw[k], rbin[k], has, check)
        assert n == ref[k][0] and np.array_equal(m12, ref[k][1]), "neighbour %d: %d vs %d matches" % (k, n, ref[k][0])
        new = np.flatnonzero(m12 >= 0)
        new = new[rng2.random(len(new)) < 0.7]
        has[new] = 1
        total += n
    assert total > 30 * min(K, 3)
    # the first neighbour == the single-pair entry point on the same data
    m = orbfe.ORBmatcher(ex)
    off1, idx1, off2, idx2 = csr(node1, nbs[0]["node"])
    n0, m0 = m.SearchForTriangulation(off1, idx1, off2, idx2, kp.view(orbfe.KP_DTYPE), desc, has1, s1, nbs[0]["kp"].view(orbfe.KP_DTYPE),
                                      nbs[0]["desc"], nbs[0]["has"], nbs[0]["stereo"], ex.mvScaleFactor, nbs[0]["F12"], nbs[0]["ep"],
                                      only_stereo, coarse, check)
    assert n0 == ref[0][0] and np.array_equal(m0, ref[0][1])


@pytest.mark.gpu
def test_batch_with_kannala_brandt_neighbours_and_edge_cases(built):
    """KannalaBrandt8 pairs (S10) through the batch kernel, a neighbour that shares no node, an empty neighbour, K == 0,
    key frames of another device / a short parameter block refused."""
    import orbfe
    from test_triangulation import KB_ARGS, _kb_frame, kb_cameras, kb_scenario
    eo, kp, desc = _kb_frame(3)
    ex = orbfe.ORBextractor(*KB_ARGS)
    cams = kb_cameras(1, False, 3)
    off1, idx1, off2, idx2, kp2, d2, h1, h2, s1, s2, F12, ep = kb_scenario(kp, desc, cams, 3, False)
    # node ids from the scenario's groups (ascending index inside a group is what the resident key frame stores)
    node1 = np.full(len(kp), -1, np.int32)
    node2 = np.full(len(kp2), -1, np.int32)
    for g in range(len(off1) - 1):
        node1[np.asarray(idx1[off1[g]:off1[g + 1]], int)] = g
        node2[np.asarray(idx2[off2[g]:off2[g + 1]], int)] = g
    o1, i1, o2, i2 = csr(node1, node2)
    n_ref, m_ref = O.search_for_triangulation(o1, i1, o2, i2, kp, desc, h1, None, kp2, d2, h2, None, eo.scaleFactors, F12, ep, False,
                                              False, True, cameras=cams)
    kf1 = orbfe.KeyFrame(ex, kp.view(orbfe.KP_DTYPE), desc, node1, ex.mvScaleFactor)
    kfa = orbfe.KeyFrame(ex, kp2.view(orbfe.KP_DTYPE), d2, node2, ex.mvScaleFactor)
    kfb = orbfe.KeyFrame(ex, kp2.view(orbfe.KP_DTYPE), d2, node2 + 100000, ex.mvScaleFactor)   # no shared node
    kfc = orbfe.KeyFrame(ex, kp2[:0].view(orbfe.KP_DTYPE), d2[:0], node2[:0], ex.mvScaleFactor)  # no features
    P = orbfe.tri_params(F12, ep, False, False, True, cams)
    raw, rbin = orbfe.SearchForTriangulation_batch(ex, kf1, h1, [kfb, kfa, kfc], [h2, h2, h2[:0]], [P, P, P])
    assert (raw[0] == -1).all() and (raw[2] == -1).all()
    n, m12 = orbfe.triangulation_select(raw[1], rbin[1], h1, True)
    assert n == n_ref and np.array_equal(m12, m_ref) and n_ref > 10
    raw0, _ = orbfe.SearchForTriangulation_batch(ex, kf1, h1, [], [], [])
    assert raw0.shape[0] == 0
    P.struct_size -= 8
    with pytest.raises(orbfe.OrbfeError):
        orbfe.SearchForTriangulation_batch(ex, kf1, h1, [kfa], [h2], [P])
    bad = kp2.copy()
    bad["octave"][0] = 8  # outside mvScaleFactors: refused when the key frame is created
    with pytest.raises(orbfe.OrbfeError):
        orbfe.KeyFrame(ex, bad.view(orbfe.KP_DTYPE), d2, node2, ex.mvScaleFactor)


def test_select_is_the_reference_tail_on_the_host(built):
    """orbfe_triangulation_select alone (no GPU): skipping + rotation histogram + ComputeThreeMaxima == the oracle's own
    tail, checked by feeding it the oracle's unfiltered matches."""
    import orbfe
    eo, kp, desc = _frame(2)
    kp, desc = kp[:500], desc[:500]
    node1 = nodes_of(kp)
    nb = neighbour(kp, desc, 77, True, False)
    has1 = (np.random.default_rng(3).random(len(kp)) < 0.2).astype(np.uint8)
    off1, idx1, off2, idx2 = csr(node1, nb["node"])
    n_raw, raw = O.search_for_triangulation(off1, idx1, off2, idx2, kp, desc, np.zeros(len(kp), np.uint8), None, nb["kp"], nb["desc"],
                                            nb["has"], None, eo.scaleFactors, nb["F12"], nb["ep"], False, False, False)
    f32 = np.float32
    rot = (kp["angle"] - nb["kp"]["angle"][np.maximum(raw, 0)]).astype(f32)
    rot = np.where(rot < 0, (rot + f32(360)).astype(f32), rot)
    b = np.floor((rot * f32(1.0 / 30)).astype(f32).astype(np.float64) + 0.5).astype(np.int64)  # roundf, non-negative values
    rbin = np.where(b == 30, 0, b).astype(np.uint8)
    for check in (False, True):
        n_ref, m_ref = O.search_for_triangulation(off1, idx1, off2, idx2, kp, desc, has1, None, nb["kp"], nb["desc"], nb["has"], None,
                                                  eo.scaleFactors, nb["F12"], nb["ep"], False, False, check)
        n, m12 = orbfe.triangulation_select(raw, rbin, has1, check)
        assert n == n_ref and np.array_equal(m12, m_ref) and n_ref > 30
