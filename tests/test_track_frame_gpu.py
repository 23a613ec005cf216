"""orbfe_track_frame: the tracking thread's per-frame chain -- ExtractORB -> isInFrustum over the local map ->
SearchByProjection (src/Tracking.cc:152-173,1059-1115; src/Frame.cc:178-189,272-333) -- as ONE submission (one captured
hipGraph), against the oracle's chain O.Extractor.extract -> O.is_in_frustum -> O.search_by_projection, bit for bit:
keypoints, descriptors, per-level counts, the records isInFrustum writes, mTrackProjXR, match indices and the count."""
import ctypes as C

import numpy as np
import pytest

import frustum_scenarios as FS
import oracle_py as O
from test_device_entry_points_gpu import _world_points_on_keypoints
from test_frustum import ON, PN

pytestmark = pytest.mark.gpu

C1 = (1000, 40000, 1.2, 8, 20, 7, 752, 480)
GRID = (64, 48)


def oracle_chain(eo, img, Fo, pts, mpd, th, nn, grid=GRID, far=False, th_far=0.0):
    W, H = eo_size(eo)
    kp, desc, per = eo.extract(img)
    mps, xr = O.is_in_frustum(Fo, pts)
    if len(kp) == 0:
        return dict(kp=kp, desc=desc, per=per, mps=mps, xr=xr, n=0, match=np.zeros(0, np.int32))
    fv = O.make_frame_view(kp, desc, grid[0], grid[1], 0.0, 0.0, float(W), float(H), eo.scaleFactors)
    n, match = O.search_by_projection(fv, mps, mpd, None, th, nn, far, th_far)
    return dict(kp=kp, desc=desc, per=per, mps=mps, xr=xr, n=n, match=match)


def eo_size(eo):
    return eo.W, eo.H


def same(got, ref, what):
    assert len(got["kp"]) == len(ref["kp"]), "%s: keypoint count %d vs %d" % (what, len(got["kp"]), len(ref["kp"]))
    assert got["kp"].tobytes() == ref["kp"].tobytes(), what + ": keypoints"
    assert np.array_equal(got["desc"], ref["desc"]), what + ": descriptors"
    assert np.array_equal(got["per_level"], ref["per"]), what + ": per-level counts"
    assert got["mps"].tobytes() == ref["mps"].tobytes(), what + ": isInFrustum records"
    assert got["proj_xr"].tobytes() == ref["xr"].tobytes(), what + ": mTrackProjXR"
    assert got["nmatches"] == ref["n"], "%s: match count %d vs %d" % (what, got["nmatches"], ref["n"])
    assert np.array_equal(got["match"], ref["match"]), what + ": match indices"


def make_oracle(args):
    return O.Extractor(*args)


def frusta(seed, W, H, n_levels=8, kb8=False):
    Fo, Fp = O.Frustum(), __import__("orbfe").Frustum()
    v = FS.fill_frustum(Fo, ON, W=float(W), H=float(H), n_levels=n_levels, seed=seed, kb8=kb8)
    FS.fill_frustum(Fp, PN, W=float(W), H=float(H), n_levels=n_levels, seed=seed, kb8=kb8)
    return Fo, Fp, v


def test_track_frame_equals_oracle_chain_on_a_stream(built):
    """C1 geometry; consecutive frames with a new pose and a local map of changing size (same bucket, next bucket, a
    smaller one again, none at all), both call parameter sets of Tracking::SearchLocalPoints, pageable / pinned /
    padded-pitch sources.  Every call replays or captures a graph; all must equal the oracle."""
    import torch
    import orbfe
    from orbfe import synth
    W, H = C1[6], C1[7]
    eo = make_oracle(C1)
    ex = orbfe.ORBextractor(*C1)
    trk = orbfe.FrameTracker(ex, GRID[0], GRID[1], 0.0, 0.0, float(W), float(H))
    frames = list(synth.stream(W, H, 9, index0=40))
    sizes = [2000, 2010, 1990, 2300, 700, 0, 2000, 5000, 1]
    total = 0
    for i, (img, M) in enumerate(zip(frames, sizes)):
        Fo, Fp, v = frusta(100 + i, W, H, kb8=(i in (3, 6)))  # two frames through the KannalaBrandt8 projection (S5 / S8 sequences)
        kp_r, desc_r, _ = eo.extract(img)
        pts, mpd = _world_points_on_keypoints(kp_r, desc_r, v, max(M, 1), np.random.default_rng(i), 8)
        pts, mpd = pts[:M], mpd[:M]
        th, nn = ((20.0, 0.85), (40.0, 0.75))[i % 2]  # before / after IMU initialisation (src/Tracking.cc:1108-1113)
        ref = oracle_chain(eo, img, Fo, pts, mpd, th, nn)
        src = img
        if i % 3 == 1:
            src = torch.from_numpy(img.copy()).pin_memory().numpy()
        elif i % 3 == 2:
            padded = torch.zeros((H, W + 16), dtype=torch.uint8).pin_memory().numpy()
            padded[:, :W] = img
            src = padded[:, :W]
        got = trk.TrackFrame(src, Fp, pts.view(orbfe.WP_DTYPE), mpd, th, nn)
        same(got, ref, "frame %d (M = %d)" % (i, M))
        total += ref["n"]
    assert total > 2000  # the chain really matches: thousands of accepted map points over the stream


def test_track_frame_equals_the_three_calls_and_the_plain_launch_path(built):
    """== orbfe_extract + orbfe_project_map_points + orbfe_match_projection on the same handle, interleaved with them (the
    handle's scratch is shared), and == its own plain-launch path (stage timing on)."""
    import orbfe
    from orbfe import synth
    args = (800, 30000, 1.2, 6, 20, 7, 640, 400)
    W, H = args[6], args[7]
    ex = orbfe.ORBextractor(*args)
    m = orbfe.ORBmatcher(ex)
    trk = orbfe.FrameTracker(ex, 40, 25, 0.0, 0.0, float(W), float(H))
    eo = make_oracle(args)
    for i in range(4):
        img = synth.frame(W, H, 900 + i)
        _, Fp, v = frusta(7 + i, W, H, 6)
        kp0, desc0 = ex.extractFeatures(img)
        pts, mpd = _world_points_on_keypoints(kp0.view(O.KP_DTYPE), desc0, v, 1500 + 200 * i, np.random.default_rng(i), 6)
        pts = pts.view(orbfe.WP_DTYPE)
        mps, xr = m.isInFrustum_batch(Fp, pts)
        fv = orbfe.make_frame_view(kp0, desc0, 40, 25, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
        n3, match3 = m.SearchByProjection(fv, mps, mpd, 20.0, False, 0.0, 0.85, None)
        got = trk.TrackFrame(img, Fp, pts, mpd, 20.0, 0.85)
        assert got["kp"].tobytes() == kp0.tobytes() and np.array_equal(got["desc"], desc0)
        assert got["mps"].tobytes() == mps.tobytes() and got["proj_xr"].tobytes() == xr.tobytes()
        assert got["nmatches"] == n3 and np.array_equal(got["match"], match3) and n3 > 100
        ex.set_stage_timing(True)  # plain launches
        plain = trk.TrackFrame(img, Fp, pts, mpd, 20.0, 0.85)
        ex.set_stage_timing(False)
        for k in ("kp", "desc", "mps", "proj_xr", "match"):
            assert plain[k].tobytes() == got[k].tobytes(), k
        assert plain["nmatches"] == got["nmatches"]
        # far-point filter on: another graph key
        far = trk.TrackFrame(img, Fp, pts, mpd, 20.0, 0.85, True, 4.0)
        nf, matchf = m.SearchByProjection(fv, mps, mpd, 20.0, True, 4.0, 0.85, None)
        assert far["nmatches"] == nf and np.array_equal(far["match"], matchf) and nf < n3
        ref = oracle_chain(eo, img, frusta(7 + i, W, H, 6)[0], pts.view(O.WP_DTYPE), mpd, 20.0, 0.85, (40, 25))
        same(got, ref, "frame %d" % i)


@pytest.mark.parametrize("kind", ["noise", "checker3", "plateau"])
def test_track_frame_on_hostile_frames(built, kind):
    """Hostile image classes (tests/test_hostile_gpu.py): white noise (dense candidates, look-alike descriptors by the
    hundred -> top-K overflow and exact rescans inside the fused chain), a period-3 checkerboard and a lattice of identical
    blobs (equal scores, equal descriptors)."""
    import orbfe
    from orbfe import synth
    W, H = C1[6], C1[7]
    eo = make_oracle(C1)
    ex = orbfe.ORBextractor(*C1)
    trk = orbfe.FrameTracker(ex, GRID[0], GRID[1], 0.0, 0.0, float(W), float(H))
    img = synth.hostile(kind, W, H, 3)
    Fo, Fp, v = frusta(31, W, H)
    kp_r, desc_r, _ = eo.extract(img)
    if len(kp_r):
        pts, mpd = _world_points_on_keypoints(kp_r, desc_r, v, 3000, np.random.default_rng(5), 8)
    else:
        pts, mpd = FS.world_points(3000, O.WP_DTYPE, dict(zip(O.WP_DTYPE.names, O.WP_DTYPE.names)), 5), np.zeros((3000, 32), np.uint8)
    ref = oracle_chain(eo, img, Fo, pts, mpd, 40.0, 0.75)
    got = trk.TrackFrame(img, Fp, pts.view(orbfe.WP_DTYPE), mpd, 40.0, 0.75)
    same(got, ref, kind)


def test_track_frame_without_keypoints_and_argument_checks(built):
    import orbfe
    args = (500, 8000, 1.2, 4, 20, 7, 320, 240)
    W, H = args[6], args[7]
    ex = orbfe.ORBextractor(*args)
    trk = orbfe.FrameTracker(ex, 16, 12, 0.0, 0.0, float(W), float(H))
    Fo, Fp, v = frusta(3, W, H, 4)
    names = dict(min_distance="minDistance", max_distance="maxDistance")
    pts = FS.world_points(900, O.WP_DTYPE, {k: names.get(k, k) for k in ("x", "y", "z", "min_distance", "max_distance", "bad", "observations", "skip")}, 4)
    mpd = np.random.default_rng(1).integers(0, 256, (900, 32), dtype=np.uint8)
    flat = np.full((H, W), 128, np.uint8)  # no corners: the reference returns before tracking (src/Tracking.cc:158-159)
    got = trk.TrackFrame(flat, Fp, pts.view(orbfe.WP_DTYPE), mpd, 20.0, 0.85)
    assert len(got["kp"]) == 0 and got["nmatches"] == 0 and len(got["match"]) == 0
    ref_mps, ref_xr = O.is_in_frustum(Fo, pts)
    assert got["mps"].tobytes() == ref_mps.tobytes() and got["proj_xr"].tobytes() == ref_xr.tobytes()
    # refused: a parameter block of another size, a frustum with more levels than the extractor, a bad camera model
    tp = orbfe.TrackParams()
    tp.grid_cols, tp.grid_rows, tp.grid_inv_w, tp.grid_inv_h, tp.th, tp.nn_ratio = 16, 12, 0.05, 0.05, 20.0, 0.85
    kp = np.zeros(ex.cap, orbfe.KP_DTYPE)
    desc = np.zeros((ex.cap, 32), np.uint8)
    match = np.zeros(ex.cap, np.int32)
    n, nm = C.c_int(), C.c_int()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    call = lambda: ex.L.orbfe_track_frame(ex.h, p(flat), W, C.byref(Fp), C.byref(tp), 900, p(pts), p(mpd), p(kp), p(desc),
                                          C.byref(n), None, None, None, p(match), C.byref(nm))
    assert call() == 0
    tp.struct_size -= 4
    assert call() == 1 and b"struct_size" in ex.L.orbfe_last_error(ex.h)
    tp.struct_size += 4
    Fp.n_levels = 5
    assert call() == 1
    Fp.n_levels = 4
    Fp.camera_model = 9
    assert call() == 2
    Fp.camera_model = 0
    assert call() == 0
