"""orbfe_track_reference_keyframe: the tracking thread's chain of a frame tracked against its reference key frame --
ExtractORB -> the per-feature part of Frame::ComputeBoW -> ORBmatcher::SearchByBoW(mpReferenceKF, mCurrentFrame, ...)
(Tracking::TrackReferenceKeyFrame, src/Tracking.cc:825-835; every frame while the IMU is not initialised, :454-458) -- as
ONE submission against a key frame resident in HBM, compared bit for bit with the oracle's chain
O.Extractor.extract -> O.vocab_transform -> (merge-walk of the two FeatureVectors) -> O.search_by_bow."""
import numpy as np
import pytest

import oracle_py as O
import vocab_synth as vs

pytestmark = pytest.mark.gpu

C1 = (1000, 40000, 1.2, 8, 20, 7, 752, 480)


def csr(nodeKF, nodeF):
    """the lockstep walk of the two FeatureVectors (src/ORBmatcher.cc:150-165,289-300): shared nodes ascending, the
    features of a node in ascending index (the order DBoW2 stores them in)"""
    kfOff, kfIdx, fOff, fIdx = [0], [], [0], []
    for g in sorted((set(nodeKF.tolist()) & set(nodeF.tolist())) - {-1}):
        kfIdx += list(np.flatnonzero(nodeKF == g))
        fIdx += list(np.flatnonzero(nodeF == g))
        kfOff.append(len(kfIdx))
        fOff.append(len(fIdx))
    return kfOff, kfIdx, fOff, fIdx


def tree_for_images(k, L, seed):
    return vs.spread_first_level(vs.make_tree(k, L, seed=seed, early_leaf_p=0.03, dup_p=0.05), seed + 1)


def oracle_chain(eo, t, levelsup, img, kf, has, nn, check):
    kp, desc, per = eo.extract(img)
    word, node, weight = O.vocab_transform(t["childOff"], t["childIdx"], t["nodeDesc"], t["wordId"], t["weight"], t["L"], desc, levelsup)
    if len(kp) == 0 or len(kf["kp"]) == 0:
        return dict(kp=kp, desc=desc, per=per, word=word, node=node, weight=weight, n=0, match=np.full(len(kp), -1, np.int32))
    # a feature enters mFeatVec only when its word's weight is > 0 (TemplatedVocabulary.h:1168-1172,1196-1200)
    kfOff, kfIdx, fOff, fIdx = csr(kf["node"], np.where(weight > 0, node, -1))
    n, match = O.search_by_bow(kfOff, kfIdx, fOff, fIdx, kf["desc"], kf["kp"]["angle"], has, desc, kp["angle"], nn, check)
    return dict(kp=kp, desc=desc, per=per, word=word, node=node, weight=weight, n=n, match=match)


def same(got, ref, what):
    assert len(got["kp"]) == len(ref["kp"]), "%s: keypoint count %d vs %d" % (what, len(got["kp"]), len(ref["kp"]))
    assert got["kp"].tobytes() == ref["kp"].tobytes(), what + ": keypoints"
    assert np.array_equal(got["desc"], ref["desc"]), what + ": descriptors"
    assert np.array_equal(got["per_level"], ref["per"]), what + ": per-level counts"
    assert np.array_equal(got["word"], ref["word"]), what + ": word ids"
    assert np.array_equal(got["node"], ref["node"]), what + ": node ids"
    assert np.array_equal(got["weight"].view(np.uint64), ref["weight"].view(np.uint64)), what + ": word weights"
    assert got["nmatches"] == ref["n"], "%s: match count %d vs %d" % (what, got["nmatches"], ref["n"])
    assert np.array_equal(got["match"], ref["match"]), what + ": match indices"


def make_kf(eo, t, levelsup, img):
    kp, desc, _ = eo.extract(img)
    _, node, weight = O.vocab_transform(t["childOff"], t["childIdx"], t["nodeDesc"], t["wordId"], t["weight"], t["L"], desc, levelsup)
    # the key frame's mFeatVec: features on stopped words (weight 0) are in no node (-1 for orbfe_keyframe_create)
    return dict(kp=kp, desc=desc, node=np.where(weight > 0, node, -1).astype(np.int32))


@pytest.mark.parametrize("k,L,levelsup", [(10, 5, 3), (10, 6, 4), (6, 4, 1), (6, 4, 4)])
def test_chain_equals_oracle_on_a_stream(built, k, L, levelsup):
    """consecutive frames of one stream against a reference key frame that changes twice on the way (the graph of the chain
    does not depend on the key frame), flags that change from frame to frame, pageable / pinned / padded sources, both
    settings of checkOrientation, two ratios.  (6, 4, 4): levelsup == L, the whole frame in one node -- the matcher's
    general walk; (6, 4, 1): ~200 nodes of a few features; the others ~100 nodes, some of them with more than 64 features."""
    import torch
    import orbfe
    from orbfe import synth
    W, H = C1[6], C1[7]
    eo = O.Extractor(*C1)
    ex = orbfe.ORBextractor(*C1)
    t = tree_for_images(k, L, seed=3 * k + L)
    voc = orbfe.ORBVocabulary(ex, t["childOff"], t["childIdx"], t["nodeDesc"], t["wordId"], t["weight"], L)
    trk = orbfe.FrameTracker(ex, 64, 48, 0.0, 0.0, float(W), float(H))
    frames = list(synth.stream(W, H, 10, index0=300))
    rng = np.random.default_rng(k)
    kf, res, total = None, None, 0
    for i, img in enumerate(frames):
        if i in (0, 4, 7):  # a new reference key frame (src/Tracking.cc:1236 CreateNewKeyFrame -> mpReferenceKF)
            kf = make_kf(eo, t, levelsup, frames[max(i - 1, 0)])
            res = orbfe.KeyFrame(ex, kf["kp"].view(orbfe.KP_DTYPE), kf["desc"], kf["node"], eo.scaleFactors)
        has = (rng.random(len(kf["kp"])) < 0.7).astype(np.uint8)
        nn, check = ((0.75, True), (0.9, True), (0.75, False))[i % 3]
        src = img
        if i % 3 == 1:
            src = torch.from_numpy(img.copy()).pin_memory().numpy()
        elif i % 3 == 2:
            padded = torch.zeros((H, W + 16), dtype=torch.uint8).pin_memory().numpy()
            padded[:, :W] = img
            src = padded[:, :W]
        got = trk.TrackReferenceKeyFrame(src, voc, levelsup, res, has, nn, check)
        ref = oracle_chain(eo, t, levelsup, img, kf, has, nn, check)
        same(got, ref, "frame %d" % i)
        total += ref["n"]
    assert total > 1000  # the chain really matches: the consecutive frames share most of their features
    voc.close()


def test_stopped_words_take_no_part_in_the_matching(built):
    """DBoW2 adds a feature to the FeatureVector only when its word's weight is > 0 (TemplatedVocabulary.h:1168-1172,
    1196-1200): with half of the words stopped, the features on them must stay unmatched -- on the frame side (decided on
    the device from the leaf's weight) and on the key-frame side (node -1) -- although they would match if they took part."""
    import orbfe
    from orbfe import synth
    W, H = C1[6], C1[7]
    eo = O.Extractor(*C1)
    ex = orbfe.ORBextractor(*C1)
    t = vs.spread_first_level(vs.make_tree(10, 5, seed=41, early_leaf_p=0.03, dup_p=0.05, stop_p=0.5), 42)
    voc = orbfe.ORBVocabulary(ex, t["childOff"], t["childIdx"], t["nodeDesc"], t["wordId"], t["weight"], 5)
    trk = orbfe.FrameTracker(ex, 64, 48, 0.0, 0.0, float(W), float(H))
    img = synth.frame(W, H, 640)
    kf = make_kf(eo, t, 3, img)  # the frame's own image as the key frame: every feature would find itself
    res = orbfe.KeyFrame(ex, kf["kp"].view(orbfe.KP_DTYPE), kf["desc"], kf["node"], eo.scaleFactors)
    has = np.ones(len(kf["kp"]), np.uint8)
    got = trk.TrackReferenceKeyFrame(img, voc, 3, res, has, 0.75, False)
    ref = oracle_chain(eo, t, 3, img, kf, has, 0.75, False)
    same(got, ref, "half of the words stopped")
    stopped = ref["weight"] <= 0
    assert 0.3 < stopped.mean() < 0.7 and (got["match"][stopped] == -1).all() and got["nmatches"] > 200
    # the unfiltered walk (every feature in its node, weight ignored) matches stopped features too: the filter matters here
    kfAll = dict(kf, node=ref["node"].astype(np.int32))
    n_all, m_all = O.search_by_bow(*csr(kfAll["node"], ref["node"]), kf["desc"], kf["kp"]["angle"], has, ref["desc"], ref["kp"]["angle"],
                                   0.75, False)
    assert (m_all[stopped] >= 0).sum() > 100 and n_all > got["nmatches"]
    voc.close()


def test_chain_equals_the_three_calls_and_the_plain_launch_path(built):
    """== orbfe_extract + orbfe_bow_transform + orbfe_match_bow on the same handle, interleaved with them, and == its own
    plain-launch path (stage timing on)."""
    import orbfe
    from orbfe import synth
    args = (800, 30000, 1.2, 6, 20, 7, 640, 400)
    W, H = args[6], args[7]
    ex = orbfe.ORBextractor(*args)
    m = orbfe.ORBmatcher(ex)
    eo = O.Extractor(*args)
    t = tree_for_images(9, 5, seed=11)
    voc = orbfe.ORBVocabulary(ex, t["childOff"], t["childIdx"], t["nodeDesc"], t["wordId"], t["weight"], 5)
    trk = orbfe.FrameTracker(ex, 40, 25, 0.0, 0.0, float(W), float(H))
    kf = make_kf(eo, t, 3, synth.frame(W, H, 500))
    res = orbfe.KeyFrame(ex, kf["kp"].view(orbfe.KP_DTYPE), kf["desc"], kf["node"], eo.scaleFactors)
    has = np.ones(len(kf["kp"]), np.uint8)
    for i in range(3):
        img = synth.frame(W, H, 500 + i)
        kp0, desc0 = ex.extractFeatures(img)
        w0, n0, wt0 = voc.transform(desc0, 3)
        kfOff, kfIdx, fOff, fIdx = csr(kf["node"], np.where(wt0 > 0, n0, -1))  # mFeatVec holds features with weight > 0 only
        n3, match3 = m.SearchByBoW(kfOff, kfIdx, fOff, fIdx, kf["desc"], kf["kp"]["angle"], has, desc0, kp0["angle"], 0.75, True)
        got = trk.TrackReferenceKeyFrame(img, voc, 3, res, has, 0.75, True)
        assert got["kp"].tobytes() == kp0.tobytes() and np.array_equal(got["desc"], desc0)
        assert np.array_equal(got["word"], w0) and np.array_equal(got["node"], n0) and np.array_equal(got["weight"], wt0)
        assert got["nmatches"] == n3 and np.array_equal(got["match"], match3)
        ex.set_stage_timing(True)  # plain launches
        plain = trk.TrackReferenceKeyFrame(img, voc, 3, res, has, 0.75, True)
        ex.set_stage_timing(False)
        for key in ("kp", "desc", "word", "node", "weight", "match"):
            assert plain[key].tobytes() == got[key].tobytes(), key
        assert plain["nmatches"] == got["nmatches"]
        if i == 0:
            assert n3 > 300  # the key frame's own image: most features find themselves
    voc.close()


def test_edge_cases(built):
    """a frame without keypoints; a key frame without features; a key frame none of whose features has a map point; a key
    frame with more features than the flag block was first sized for (the blocks regrow, the graphs are re-captured); a
    degenerate vocabulary position (levelsup >= L: every feature in the root's node -- one node holds the whole frame)."""
    import orbfe
    from orbfe import synth
    args = (500, 20000, 1.2, 4, 20, 7, 320, 240)
    W, H = args[6], args[7]
    ex = orbfe.ORBextractor(*args)
    eo = O.Extractor(*args)
    t = tree_for_images(8, 4, seed=5)
    voc = orbfe.ORBVocabulary(ex, t["childOff"], t["childIdx"], t["nodeDesc"], t["wordId"], t["weight"], 4)
    trk = orbfe.FrameTracker(ex, 16, 12, 0.0, 0.0, float(W), float(H))
    img = synth.frame(W, H, 77)
    kf = make_kf(eo, t, 2, synth.frame(W, H, 76))
    res = orbfe.KeyFrame(ex, kf["kp"].view(orbfe.KP_DTYPE), kf["desc"], kf["node"], eo.scaleFactors)
    has = np.ones(len(kf["kp"]), np.uint8)
    # blank frame: no keypoints, no matches
    got = trk.TrackReferenceKeyFrame(np.full((H, W), 90, np.uint8), voc, 2, res, has)
    assert len(got["kp"]) == 0 and got["nmatches"] == 0 and len(got["match"]) == 0
    # the same handle afterwards
    same(trk.TrackReferenceKeyFrame(img, voc, 2, res, has), oracle_chain(eo, t, 2, img, kf, has, 0.75, True), "after blank")
    # no flags set
    none = np.zeros(len(kf["kp"]), np.uint8)
    got = trk.TrackReferenceKeyFrame(img, voc, 2, res, none)
    assert got["nmatches"] == 0 and (got["match"] == -1).all()
    # empty key frame
    empty = orbfe.KeyFrame(ex, np.zeros(0, orbfe.KP_DTYPE), np.zeros((0, 32), np.uint8), np.zeros(0, np.int32), eo.scaleFactors)
    got = trk.TrackReferenceKeyFrame(img, voc, 2, empty, np.zeros(0, np.uint8))
    assert got["nmatches"] == 0 and len(got["kp"]) > 100 and (got["match"] == -1).all()
    # a big key frame (a stereo key frame of two 3000-feature images, say): 6000 flags > the first block
    rng = np.random.default_rng(9)
    reps = 6000 // len(kf["kp"]) + 1
    big = dict(kp=np.tile(kf["kp"], reps)[:6000].copy(), desc=np.tile(kf["desc"], (reps, 1))[:6000].copy(),
               node=np.tile(kf["node"], reps)[:6000].copy())
    flip = rng.random(6000) < 0.5
    big["desc"][flip, 0] ^= rng.integers(1, 256, int(flip.sum()), dtype=np.uint8)
    resb = orbfe.KeyFrame(ex, big["kp"].view(orbfe.KP_DTYPE), big["desc"], big["node"], eo.scaleFactors)
    hasb = (rng.random(6000) < 0.6).astype(np.uint8)
    same(trk.TrackReferenceKeyFrame(img, voc, 2, resb, hasb, 0.8, True), oracle_chain(eo, t, 2, img, big, hasb, 0.8, True), "big key frame")
    same(trk.TrackReferenceKeyFrame(img, voc, 2, res, has), oracle_chain(eo, t, 2, img, kf, has, 0.75, True), "small one again")
    # levelsup >= L: nid level <= 0 -> every feature's node is the root
    kf0 = make_kf(eo, t, 4, synth.frame(W, H, 76))
    assert ((kf0["node"] == 0) | (kf0["node"] == -1)).all() and (kf0["node"] == 0).sum() > 100  # (-1: stopped words)
    res0 = orbfe.KeyFrame(ex, kf0["kp"].view(orbfe.KP_DTYPE), kf0["desc"], kf0["node"], eo.scaleFactors)
    same(trk.TrackReferenceKeyFrame(img, voc, 4, res0, has), oracle_chain(eo, t, 4, img, kf0, has, 0.75, True), "one node")
    # invalid arguments are refused
    with pytest.raises(orbfe.OrbfeError):
        ex._chk(ex.L.orbfe_track_reference_keyframe(ex.h, None, W, voc.v, 2, res.h, None, 0.75, 1, None, None, None, None, None, None,
                                                    None, None, None), "orbfe_track_reference_keyframe")
    voc.close()
