"""orbfe_track_initialization: the tracking thread's chain of a frame while the map is being initialised --
ExtractORB -> ORBmatcher::SearchForInitialization(mInitialFrame, mCurrentFrame, 40, 0.45, true)
(Tracking::MonocularInitialization, src/Tracking.cc:566-607; src/ORBmatcher.cc:329-439) -- as ONE submission against an
initial frame resident in HBM, compared bit for bit with the oracle's O.Extractor.extract + O.search_for_initialization."""
import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu

C1 = (1000, 40000, 1.2, 8, 20, 7, 752, 480)
FEAT_INIT_COUNT = 100


def oracle_chain(eo, kp1, d1, img, grid, window, nn, check):
    W, H = eo_wh(eo)
    kp, desc, per = eo.extract(img)
    fv1 = O.make_frame_view(kp1, d1, grid[0], grid[1], 0.0, 0.0, float(W), float(H), eo.scaleFactors)
    fv2 = O.make_frame_view(kp, desc, grid[0], grid[1], 0.0, 0.0, float(W), float(H), eo.scaleFactors)
    n, m12 = O.search_for_initialization(fv1, fv2, window, nn, check)
    return dict(kp=kp, desc=desc, per=per, n=n, m12=m12)


def eo_wh(eo):
    return eo.W, eo.H


def same(got, ref, what):
    assert len(got["kp"]) == len(ref["kp"]), "%s: keypoint count %d vs %d" % (what, len(got["kp"]), len(ref["kp"]))
    assert got["kp"].tobytes() == ref["kp"].tobytes(), what + ": keypoints"
    assert np.array_equal(got["desc"], ref["desc"]), what + ": descriptors"
    assert np.array_equal(got["per_level"], ref["per"]), what + ": per-level counts"
    assert got["nmatches"] == ref["n"], "%s: match count %d vs %d" % (what, got["nmatches"], ref["n"])
    assert np.array_equal(got["matches12"], ref["m12"]), what + ": vnMatches12"


def test_chain_equals_oracle_on_a_stream_with_the_initial_frame_replaced_twice(built):
    """A stream as the tracking thread sees it: the first frame with more than FEAT_INIT_COUNT keypoints becomes mInitialFrame
    (:569-586); every later frame is extracted and matched against it; the attempt is reset and re-seeded twice on the way
    (:588-602: time-out / too few keypoints), so three different resident initial frames serve one handle.  Pageable, pinned
    and padded sources; the reference's own parameters (40, 0.45, true) and two other sets."""
    import torch
    import orbfe
    from orbfe import synth
    W, H = C1[6], C1[7]
    eo = O.Extractor(*C1)
    eo.W, eo.H = W, H
    ex = orbfe.ORBextractor(*C1)
    trk = orbfe.FrameTracker(ex, 64, 48, 0.0, 0.0, float(W), float(H))
    frames = list(synth.stream(W, H, 12, index0=500))
    ini, kp1, d1, total = None, None, None, 0
    for i, img in enumerate(frames):
        if i in (0, 5, 9):  # (re-)seed: this frame becomes the initial frame
            kp1, d1, _ = eo.extract(img)
            assert len(kp1) > FEAT_INIT_COUNT
            if ini is not None:
                ini.close()
            ini = orbfe.InitialFrame(ex, kp1.view(orbfe.KP_DTYPE), d1)
            continue
        window, nn, check = ((40, 0.45, True), (100, 0.9, True), (40, 0.45, False))[i % 3]
        src = img
        if i % 3 == 1:
            src = torch.from_numpy(img.copy()).pin_memory().numpy()
        elif i % 3 == 2:
            padded = torch.zeros((H, W + 16), dtype=torch.uint8).pin_memory().numpy()
            padded[:, :W] = img
            src = padded[:, :W]
        got = trk.TrackInitialization(src, ini, window, nn, check)
        ref = oracle_chain(eo, kp1, d1, img, (64, 48), window, nn, check)
        same(got, ref, "frame %d" % i)
        total += ref["n"]
    assert total > 300  # consecutive frames of one scene: the chain really matches
    captured, failed = ex.graph_stats()
    assert failed == 0 and captured >= 3
    ini.close()


def test_chain_equals_the_two_calls_and_the_plain_launch_path(built):
    """== orbfe_extract + orbfe_match_initialization on the same handle, interleaved with them, and == its own plain-launch path
    (stage timing on)."""
    import orbfe
    from orbfe import synth
    args = (800, 30000, 1.2, 6, 20, 7, 640, 400)
    W, H = args[6], args[7]
    ex = orbfe.ORBextractor(*args)
    m = orbfe.ORBmatcher(ex)
    trk = orbfe.FrameTracker(ex, 40, 25, 0.0, 0.0, float(W), float(H))
    kp1, d1 = ex.extractFeatures(synth.frame(W, H, 700))
    ini = orbfe.InitialFrame(ex, kp1, d1)
    fv1 = orbfe.make_frame_view(kp1, d1, 40, 25, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
    for i in range(3):
        img = synth.frame(W, H, 700 + i)
        kp2, d2 = ex.extractFeatures(img)
        fv2 = orbfe.make_frame_view(kp2, d2, 40, 25, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
        n2, m2 = m.SearchForInitialization(fv1, fv2, 40, 0.45, True)
        got = trk.TrackInitialization(img, ini, 40, 0.45, True)
        assert got["kp"].tobytes() == kp2.tobytes() and np.array_equal(got["desc"], d2)
        assert got["nmatches"] == n2 and np.array_equal(got["matches12"], m2)
        ex.set_stage_timing(True)  # plain launches
        plain = trk.TrackInitialization(img, ini, 40, 0.45, True)
        ex.set_stage_timing(False)
        for key in ("kp", "desc", "matches12"):
            assert plain[key].tobytes() == got[key].tobytes(), key
        assert plain["nmatches"] == got["nmatches"]
        if i == 0:
            assert n2 > 100  # the initial frame's own image: its level-0 features find themselves
    ini.close()


def test_edge_cases(built):
    """a current frame without keypoints; an initial frame without keypoints; an initial frame without level-0 keypoints; a
    configuration whose frames exceed the matcher's LDS image (the sequential kernel with frame 2 in global memory)."""
    import orbfe
    from orbfe import synth
    args = (500, 20000, 1.2, 4, 20, 7, 320, 240)
    W, H = args[6], args[7]
    ex = orbfe.ORBextractor(*args)
    eo = O.Extractor(*args)
    eo.W, eo.H = W, H
    trk = orbfe.FrameTracker(ex, 16, 12, 0.0, 0.0, float(W), float(H))
    img = synth.frame(W, H, 77)
    kp1, d1, _ = eo.extract(synth.frame(W, H, 76))
    ini = orbfe.InitialFrame(ex, kp1.view(orbfe.KP_DTYPE), d1)
    got = trk.TrackInitialization(np.full((H, W), 90, np.uint8), ini)  # blank frame: no keypoints, no matches
    assert len(got["kp"]) == 0 and got["nmatches"] == 0 and (got["matches12"] == -1).all() and len(got["matches12"]) == len(kp1)
    same(trk.TrackInitialization(img, ini), oracle_chain(eo, kp1, d1, img, (16, 12), 40, 0.45, True), "after blank")
    empty = orbfe.InitialFrame(ex, np.zeros(0, orbfe.KP_DTYPE), np.zeros((0, 32), np.uint8))
    got = trk.TrackInitialization(img, empty)
    assert got["nmatches"] == 0 and len(got["matches12"]) == 0 and len(got["kp"]) > 100
    upper = kp1[kp1["octave"] > 0]  # no level-0 keypoint: SearchForInitialization skips every feature (:346-347)
    ini_u = orbfe.InitialFrame(ex, upper.view(orbfe.KP_DTYPE), d1[kp1["octave"] > 0])
    got = trk.TrackInitialization(img, ini_u)
    assert got["nmatches"] == 0 and (got["matches12"] == -1).all() and len(got["matches12"]) == len(upper)
    same(trk.TrackInitialization(img, ini), oracle_chain(eo, kp1, d1, img, (16, 12), 40, 0.45, True), "first one again")
    for f in (ini, empty, ini_u):
        f.close()
    # 3000 features: more than 2048 keypoints per frame -> frame 2 does not fit the LDS image of the matcher
    big = (3000, 60000, 1.2, 8, 20, 7, 752, 480)
    exb = orbfe.ORBextractor(*big)
    eob = O.Extractor(*big)
    eob.W, eob.H = 752, 480
    trb = orbfe.FrameTracker(exb, 64, 48, 0.0, 0.0, 752.0, 480.0)
    fr = list(synth.stream(752, 480, 2, index0=40))
    kpb, db, _ = eob.extract(fr[0])
    assert len(kpb) > 2048
    inib = orbfe.InitialFrame(exb, kpb.view(orbfe.KP_DTYPE), db)
    same(trb.TrackInitialization(fr[1], inib, 40, 0.9, True), oracle_chain(eob, kpb, db, fr[1], (64, 48), 40, 0.9, True), "3000 features")
    inib.close()
    # invalid arguments are refused
    with pytest.raises(orbfe.OrbfeError):
        ex._chk(ex.L.orbfe_track_initialization(ex.h, None, W, None, None, 40, 0.45, 1, None, None, None, None, None, None), "null")
