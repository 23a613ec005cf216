"""orbfe_fuse_search_keyframe: the search part of ORBmatcher::Fuse(pKF, vpMapPoints, th) (src/ORBmatcher.cc:678-836) against a
key frame and map points that are RESIDENT in HBM -- what the 2 x K calls of LocalMapping::SearchInNeighbors
(src/LocalMapping.cc:764-860, calls at :822,:852) need.  Compared with the oracle's fuse_search on the same key frame, map points
and flags: the cases of tests/test_fuse.py through the new entry point, then a K = 20 neighbour sequence with Replace-style
descriptor / position updates of the map between the calls."""
import numpy as np
import pytest

import frustum_scenarios as FS
import match_scenarios as S
import oracle_py as O
import test_fuse as TF
from test_frustum import ON, PN

pytestmark = pytest.mark.gpu

W, H = TF.W, TF.H
ARGS = TF.ARGS


def resident(orbfe, ex, kp, desc, inv_s2, u_right, sf):
    kf = orbfe.KeyFrame(ex, kp.view(orbfe.KP_DTYPE), desc, np.full(len(kp), -1, np.int32), sf)
    kf.set_grid(64, 48, 0.0, 0.0, float(W), float(H), inv_s2, u_right)
    return kf


def ids_of(pts, base=0):
    """the id list of a call: entry base + i, complemented where the reference skips the point for THIS key frame"""
    ids = np.arange(base, base + len(pts), dtype=np.int32)
    return np.where(pts["skip"] != 0, ~ids, ids).astype(np.int32)


@pytest.mark.parametrize("M,th,stereo,seed,kb8", [(2000, 3.0, False, 1, False), (1500, 3.0, True, 2, False), (700, 8.0, True, 3, False),
                                                  (1, 3.0, False, 4, False), (1500, 4.0, False, 5, True)])
def test_resident_fuse_search_matches_oracle(built, M, th, stereo, seed, kb8):
    import orbfe
    from orbfe import synth
    eo = O.Extractor(*ARGS)
    kp, desc, _ = eo.extract(synth.frame(W, H, 6 + seed))
    ex = orbfe.ORBextractor(*ARGS)
    m = orbfe.ORBmatcher(ex)
    Fo, Fp = O.Frustum(), orbfe.Frustum()
    v = FS.fill_frustum(Fo, ON, seed=20 + seed, kb8=kb8)
    FS.fill_frustum(Fp, PN, seed=20 + seed, kb8=kb8)
    pts, mpd, u_right, inv_s2 = TF.scenario(kp, desc, eo.scaleFactors, v, M, seed, stereo)
    fvo = O.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, float(W), float(H), eo.scaleFactors)
    bi_r, bd_r = O.fuse_search(fvo, inv_s2, u_right, Fo, th, pts, mpd)
    kf = resident(orbfe, ex, kp, desc, inv_s2, u_right, eo.scaleFactors)
    mp = orbfe.MapPoints(ex, M + 50)
    stored = pts.copy()
    stored["skip"] = 0  # the skip flag of a resident entry is per call: it travels with the id
    mp.update(np.arange(7, 7 + M), stored.view(orbfe.WP_DTYPE), mpd)
    bi, bd = m.Fuse_search_keyframe(kf, mp, ids_of(pts, 7), Fp, th)
    assert np.array_equal(bd, bd_r) and np.array_equal(bi, bi_r)
    if M > 100 and not kb8:
        assert (bd_r <= 30).sum() > M // 8
    # == the host-pointer entry point on the same inputs
    fv = orbfe.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
    bi_h, bd_h = m.Fuse_search(fv, inv_s2, u_right, Fp, th, pts.view(orbfe.WP_DTYPE), mpd)
    assert np.array_equal(bi, bi_h) and np.array_equal(bd, bd_h)
    # ids outside the map and never-written entries are no points; an empty list is fine
    odd = np.array([M + 49, M + 1000, -(M + 1000), 3, 7], np.int32)  # unwritten, beyond, ~beyond, unwritten, first entry
    bi_o, bd_o = m.Fuse_search_keyframe(kf, mp, odd, Fp, th)
    first = pts[:1].copy()
    first["skip"] = 0
    bi_f, bd_f = O.fuse_search(fvo, inv_s2, u_right, Fo, th, first, mpd[:1])
    assert (bi_o[:4] == -1).all() and (bd_o[:4] == 256).all() and (bi_o[4], bd_o[4]) == (bi_f[0], bd_f[0])
    assert len(m.Fuse_search_keyframe(kf, mp, np.zeros(0, np.int32), Fp, th)[0]) == 0
    mp.close()
    kf.close()


def test_search_in_neighbors_sequence(built):
    """LocalMapping::SearchInNeighbors: the current key frame's map points are fused into K = 20 neighbours one after the other;
    between two calls the caller edits the map (Fuse :829-849: a matched point is REPLACED by the neighbour's -- its descriptor
    and position change -- or added to the key frame -- it is skipped from then on where IsInKeyFrame holds).  Resident key
    frames, a resident map updated through orbfe_map_update between the calls; every call equals the oracle's call on the
    state of that moment."""
    import orbfe
    from orbfe import synth
    eo = O.Extractor(*ARGS)
    ex = orbfe.ORBextractor(*ARGS)
    m = orbfe.ORBmatcher(ex)
    kp0, desc0, _ = eo.extract(synth.frame(W, H, 90))
    K, M = 20, 1200
    rng = np.random.default_rng(5)
    Fo0 = O.Frustum()
    v0 = FS.fill_frustum(Fo0, ON, seed=60)
    pts, mpd, _, _ = TF.scenario(kp0, desc0, eo.scaleFactors, v0, M, 3, False)
    pts["skip"] = 0
    mp = orbfe.MapPoints(ex, M)
    mp.update(np.arange(M), pts.view(orbfe.WP_DTYPE), mpd)
    neighbours = []
    for k in range(K):  # neighbour key frames: the same scene seen again -- features shuffled, moved by a fraction of a pixel,
        perm = rng.permutation(len(kp0))[:len(kp0) - 10 * k]  # a few bits of every descriptor flipped, some features missing --
        kpk = kp0[perm].copy()                                 # under the pose the points were made for; mono or stereo
        kpk["x"] += rng.normal(0, 0.3, len(kpk)).astype(np.float32)
        kpk["y"] += rng.normal(0, 0.3, len(kpk)).astype(np.float32)
        desck = np.stack([S.flip_bits(desc0[i], int(rng.integers(0, 12)), rng) for i in perm])
        Fo, Fp = O.Frustum(), orbfe.Frustum()
        FS.fill_frustum(Fo, ON, seed=60)
        FS.fill_frustum(Fp, PN, seed=60)
        u_right = None
        if k % 4 == 1:
            u_right = np.where(rng.random(len(kpk)) < 0.5, kpk["x"] - rng.uniform(2, 30, len(kpk)), -1.0).astype(np.float32)
        inv_s2 = (1.0 / (eo.scaleFactors.astype(np.float32) ** 2)).astype(np.float32)
        neighbours.append(dict(kp=kpk, desc=desck, Fo=Fo, Fp=Fp, ur=u_right, is2=inv_s2,
                               kf=resident(orbfe, ex, kpk, desck, inv_s2, u_right, eo.scaleFactors)))
    total = 0
    for k, nb in enumerate(neighbours):
        in_kf = rng.random(M) < 0.1 + 0.02 * k  # pMP->IsInKeyFrame(pKF) for this neighbour
        call_pts = pts.copy()
        call_pts["skip"] = in_kf
        ids = np.where(in_kf, ~np.arange(M, dtype=np.int32), np.arange(M, dtype=np.int32)).astype(np.int32)
        fvo = O.make_frame_view(nb["kp"], nb["desc"], 64, 48, 0.0, 0.0, float(W), float(H), eo.scaleFactors)
        bi_r, bd_r = O.fuse_search(fvo, nb["is2"], nb["ur"], nb["Fo"], 3.0, call_pts, mpd)
        bi, bd = m.Fuse_search_keyframe(nb["kf"], mp, ids, nb["Fp"], 3.0)
        assert np.array_equal(bi, bi_r) and np.array_equal(bd, bd_r), "neighbour %d" % k
        hit = np.flatnonzero(bd_r <= orbfe.ORBmatcher.TH_LOW)
        total += len(hit)
        # Replace: about half of the matched points take over the neighbour feature's descriptor (a few bits away) and move a little
        rep = hit[rng.random(len(hit)) < 0.5]
        if len(rep):
            mpd[rep] = np.stack([S.flip_bits(nb["desc"][bi_r[i]], int(rng.integers(0, 6)), rng) for i in rep])
            pts["x"][rep] += rng.normal(0, 0.01, len(rep)).astype(np.float32)
            pts["observations"][rep] += 1
            mp.update(rep, pts[rep].view(orbfe.WP_DTYPE), mpd[rep])
    assert total > 2000  # the sequence really fuses
    for nb in neighbours:
        nb["kf"].close()
    mp.close()


def test_grid_is_required_and_can_be_replaced(built):
    import orbfe
    from orbfe import synth
    eo = O.Extractor(*ARGS)
    kp, desc, _ = eo.extract(synth.frame(W, H, 33))
    ex = orbfe.ORBextractor(*ARGS)
    m = orbfe.ORBmatcher(ex)
    Fo, Fp = O.Frustum(), orbfe.Frustum()
    v = FS.fill_frustum(Fo, ON, seed=8)
    FS.fill_frustum(Fp, PN, seed=8)
    pts, mpd, _, inv_s2 = TF.scenario(kp, desc, eo.scaleFactors, v, 500, 1, False)
    mp = orbfe.MapPoints(ex, 500)
    st = pts.copy()
    st["skip"] = 0
    mp.update(np.arange(500), st.view(orbfe.WP_DTYPE), mpd)
    kf = orbfe.KeyFrame(ex, kp.view(orbfe.KP_DTYPE), desc, np.full(len(kp), -1, np.int32), eo.scaleFactors)
    with pytest.raises(orbfe.OrbfeError) as e:
        m.Fuse_search_keyframe(kf, mp, ids_of(pts), Fp, 3.0)
    assert e.value.code == 1 and "grid" in str(e.value)
    for cols, rows in ((64, 48), (32, 20), (64, 48)):  # the reference's grid, another one, the first again
        kf.set_grid(cols, rows, 0.0, 0.0, float(W), float(H), inv_s2, None)
        fvo = O.make_frame_view(kp, desc, cols, rows, 0.0, 0.0, float(W), float(H), eo.scaleFactors)
        bi_r, bd_r = O.fuse_search(fvo, inv_s2, None, Fo, 3.0, pts, mpd)
        bi, bd = m.Fuse_search_keyframe(kf, mp, ids_of(pts), Fp, 3.0)
        assert np.array_equal(bi, bi_r) and np.array_equal(bd, bd_r), (cols, rows)
    # an empty key frame: nothing qualifies
    empty = orbfe.KeyFrame(ex, np.zeros(0, orbfe.KP_DTYPE), np.zeros((0, 32), np.uint8), np.zeros(0, np.int32), eo.scaleFactors)
    empty.set_grid(64, 48, 0.0, 0.0, float(W), float(H), inv_s2, None)
    bi, bd = m.Fuse_search_keyframe(empty, mp, ids_of(pts), Fp, 3.0)
    assert (bi == -1).all() and (bd == 256).all()
