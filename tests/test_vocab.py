"""the per-feature vocabulary-tree descent of Frame::ComputeBoW
(src/Frame.cc:483-495 -> TemplatedVocabulary::transform, TemplatedVocabulary.h:1227-1270)."""
import numpy as np
import pytest

import oracle_py as orc
import vocab_synth as vs


def py_transform(t, feat, levelsup):
    """Straight-line restatement of TemplatedVocabulary.h:1227-1270 for ONE feature (pins the C oracle)."""
    co, ci, nd = t["childOff"], t["childIdx"], t["nodeDesc"]
    nid_level = t["L"] - levelsup
    nid = 0 if nid_level <= 0 else None
    final, lvl = 0, 0
    while True:
        lvl += 1
        ch = ci[co[final]:co[final + 1]]
        final = int(ch[0])
        best = int(np.unpackbits(nd[final] ^ feat).sum())
        for c in ch[1:]:
            d = int(np.unpackbits(nd[c] ^ feat).sum())
            if d < best:
                best, final = d, int(c)
        if nid is None and lvl == nid_level:
            nid = final
        if co[final + 1] == co[final]:
            break
    if nid is None:
        nid = final  # SPEC DECISION S7
    return int(t["wordId"][final]), nid, float(t["weight"][final])


@pytest.mark.parametrize("k,L,levelsup", [(10, 4, 2), (3, 5, 4), (10, 3, 4), (7, 4, 0)])
def test_oracle_vocab_matches_restatement(k, L, levelsup):
    t = vs.make_tree(k, L, seed=k * 10 + L)
    feats = vs.features_near(t, 300, seed=5)
    w, n, wt = orc.vocab_transform(t["childOff"], t["childIdx"], t["nodeDesc"], t["wordId"], t["weight"], L, feats,
                                   levelsup)
    for i in range(len(feats)):
        assert (int(w[i]), int(n[i]), float(wt[i])) == py_transform(t, feats[i], levelsup)


def test_text_loader_roundtrip(tmp_path):
    import orbfe
    t = vs.make_tree(6, 3, seed=3)
    p = tmp_path / "voc.txt"
    vs.write_text(t, str(p), scoring=0, weighting=0)
    v = orbfe.load_vocabulary_text(str(p))
    for key in ("childOff", "childIdx", "nodeDesc", "wordId", "weight"):
        assert np.array_equal(v[key], t[key]), key
    assert (v["k"], v["L"]) == (6, 3)


def ref_bow(word, node, w, weighting, scoring):
    """TemplatedVocabulary.h:1136-1204 + BowVector::normalize from the oracle's per-feature triples."""
    bow, fv = {}, {}
    for i in range(len(word)):
        if w[i] > 0:
            k = int(word[i])
            if weighting in (0, 1):
                bow[k] = bow.get(k, 0.0) + float(w[i])
            else:
                bow.setdefault(k, float(w[i]))
            fv.setdefault(int(node[i]), []).append(i)
    bow = dict(sorted(bow.items()))
    must = scoring != 5
    if weighting in (0, 1) and bow and not must:
        for k in bow:
            bow[k] /= float(len(bow))
    if must:
        if scoring == 1:
            norm = 0.0
            for v in bow.values():
                norm += v * v
            norm = float(np.sqrt(np.float64(norm)))
        else:
            norm = 0.0
            for v in bow.values():
                norm += abs(v)
        if norm > 0:
            for k in bow:
                bow[k] /= norm
    return bow, dict(sorted(fv.items()))


@pytest.mark.gpu
@pytest.mark.parametrize("k,L,levelsup,n", [(10, 4, 2, 2000), (10, 6, 4, 1500), (3, 5, 4, 700), (10, 3, 4, 64),
                                            (20, 3, 1, 999), (7, 4, 0, 1)])
def test_gpu_vocab_transform_matches_oracle(k, L, levelsup, n):
    import orbfe
    t = vs.make_tree(k, L, seed=k + L, early_leaf_p=0.05 if L > 4 else 0.1)
    feats = vs.features_near(t, n, seed=n)
    e = orbfe.ORBextractor(500, 2000, 1.2, 4, 20, 7, 320, 240)
    voc = orbfe.ORBVocabulary(e, t["childOff"], t["childIdx"], t["nodeDesc"], t["wordId"], t["weight"], L)
    gw, gn, gwt = voc.transform(feats, levelsup)
    ow, on, owt = orc.vocab_transform(t["childOff"], t["childIdx"], t["nodeDesc"], t["wordId"], t["weight"], L, feats,
                                      levelsup)
    assert np.array_equal(gw, ow)
    assert np.array_equal(gn, on)
    assert np.array_equal(gwt.view(np.uint64), owt.view(np.uint64))
    for weighting, scoring in [(0, 0), (1, 1), (2, 0), (3, 5), (0, 5)]:
        bow, fv = voc.transform_bow(feats, levelsup, weighting, scoring)
        rb, rf = ref_bow(ow, on, owt, weighting, scoring)
        assert list(bow.items()) == list(rb.items())
        assert fv == rf
    # empty input and repeated use of the same handle
    w0, n0, wt0 = voc.transform(np.zeros((0, 32), np.uint8), levelsup)
    assert len(w0) == 0
    gw2, _, _ = voc.transform(feats[: max(1, n // 3)], levelsup)
    assert np.array_equal(gw2, ow[: max(1, n // 3)])
    voc.close()


@pytest.mark.gpu
def test_gpu_vocab_rejects_malformed_tree():
    import orbfe
    t = vs.make_tree(4, 3, seed=1)
    e = orbfe.ORBextractor(500, 2000, 1.2, 4, 20, 7, 320, 240)
    bad = t["childIdx"].copy()
    bad[5] = 0  # a child pointing back at the root: not a tree
    with pytest.raises(orbfe.OrbfeError):
        orbfe.ORBVocabulary(e, t["childOff"], bad, t["nodeDesc"], t["wordId"], t["weight"], 3)
    bad2 = t["childOff"].copy()
    bad2[3] = bad2[2] - 1
    with pytest.raises(orbfe.OrbfeError):
        orbfe.ORBVocabulary(e, bad2, t["childIdx"], t["nodeDesc"], t["wordId"], t["weight"], 3)
