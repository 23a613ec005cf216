"""The extract-and-match form of the host-pointer ring (orbfe_stream_enable_track / submit_track / collect_track) on map
points resident in HBM (orbfe_map_*): per frame ExtractORB -> isInFrustum of the frame's local map points against the
frame's own pose -> SearchByProjection (src/Tracking.cc:152-173,1059-1115), frames named by host pointers, map points by
id.  Against the oracle's chain and against orbfe_track_frame frame by frame; the ring with one producer thread and one
consumer thread (the reference's grabber thread / tracking thread split, mono_inertial_node.cpp:207-210)."""
import threading

import numpy as np
import pytest

import frustum_scenarios as FS
import oracle_py as O
from test_frustum import ON, PN

pytestmark = pytest.mark.gpu

ARGS = (600, 24000, 1.2, 6, 20, 7, 376, 240)
GRID = (32, 20)


def build_scene(n_frames, pts_per_frame, seed=0):
    """frames + a map whose points re-project onto keypoints of the frames they were made from; every frame has its own
    pose.  Returns frames, per-frame (Frustum oracle, Frustum orbfe), map arrays (WP_DTYPE oracle layout, desc)."""
    import orbfe
    from orbfe import synth
    W, H = ARGS[6], ARGS[7]
    eo = O.Extractor(*ARGS)
    frames = np.stack(list(synth.stream(W, H, n_frames, index0=900 + seed)))
    frusta_o, frusta_p, pts_all, desc_all, owner = [], [], [], [], []
    for f in range(n_frames):
        Fo, Fp = O.Frustum(), orbfe.Frustum()
        kb8 = f % 5 == 4  # every fifth frame sees the map through a KannalaBrandt8 camera: per-frame frusta really are per frame
        v = FS.fill_frustum(Fo, ON, W=float(W), H=float(H), n_levels=6, seed=50 + f, kb8=kb8)
        FS.fill_frustum(Fp, PN, W=float(W), H=float(H), n_levels=6, seed=50 + f, kb8=kb8)
        kp, desc, _ = eo.extract(frames[f])
        pts, mpd = FS.world_points_on_keypoints(kp, desc, v, pts_per_frame, np.random.default_rng(seed * 1000 + f), 6)
        pts["skip"] = 0
        frusta_o.append(Fo)
        frusta_p.append(Fp)
        pts_all.append(pts)
        desc_all.append(mpd)
        owner += [f] * pts_per_frame
    return eo, frames, frusta_o, frusta_p, np.concatenate(pts_all), np.concatenate(desc_all), np.asarray(owner)


def local_ids(f, owner, n_points, map_cap, rng):
    """frame f's local map: its own points first (shuffled in), points of other frames, a few skipped (~id) and a few
    ids outside the map"""
    own = np.flatnonzero(owner == f)
    others = rng.choice(np.flatnonzero(owner != f), n_points - len(own) - 6, replace=False)
    ids = np.concatenate([own, others]).astype(np.int32)
    rng.shuffle(ids)
    ids = np.concatenate([ids, np.array([map_cap + 5, 2 ** 30, map_cap, map_cap + 1, map_cap + 77, 2 ** 31 - 2], np.int32)])
    skip = rng.random(len(ids)) < 0.03
    return np.where(skip, ~ids, ids).astype(np.int32)


def oracle_frame(eo, img, Fo, ids, map_pts, map_desc, th, nn):
    W, H = ARGS[6], ARGS[7]
    kp, desc, per = eo.extract(img)
    raw = ids.astype(np.int64)
    idx = np.where(raw < 0, ~raw, raw)
    inside = idx < len(map_pts)
    pts = np.zeros(len(ids), O.WP_DTYPE)
    mpd = np.zeros((len(ids), 32), np.uint8)
    pts[inside] = map_pts[idx[inside]]
    mpd[inside] = map_desc[idx[inside]]
    pts["skip"] = np.where(inside, (raw < 0).astype(np.int32), 1)
    pts["bad"] = np.where(inside, pts["bad"], 1)
    mps, _ = O.is_in_frustum(Fo, pts)
    fv = O.make_frame_view(kp, desc, GRID[0], GRID[1], 0.0, 0.0, float(W), float(H), eo.scaleFactors)
    n, match = O.search_by_projection(fv, mps, mpd, None, th, nn)
    return kp, desc, per, match, n, pts, mpd


def run_ring(st, frames, frusta_p, ids_all, slot, th, nn, threaded=False):
    import orbfe
    n_frames = len(frames)
    chunks = [(lo, min(slot, n_frames - lo)) for lo in range(0, n_frames, slot)]
    got = []

    def submit_all():
        for lo, n in chunks:
            fr = (orbfe.Frustum * n)(*frusta_p[lo:lo + n])
            while not st.submit_track(frames[lo:lo + n], fr, ids_all[lo:lo + n], th, nn):
                if not threaded:
                    got.extend(st.collect_track())

    if threaded:
        errs = []

        def producer():
            try:
                submit_all()
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        t = threading.Thread(target=producer)
        t.start()
        while len(got) < n_frames:
            if st.in_flight():
                got.extend(st.collect_track())
            assert not errs, errs
        t.join()
        assert not errs, errs
    else:
        submit_all()
        while st.in_flight():
            got.extend(st.collect_track())
    return got


@pytest.mark.parametrize("threaded", [False, True])
def test_ring_track_equals_oracle_and_track_frame(built, threaded):
    import orbfe
    W, H = ARGS[6], ARGS[7]
    n_frames, slot, per_frame, n_points = 27, 8, 260, 900
    eo, frames, frusta_o, frusta_p, map_pts, map_desc, owner = build_scene(n_frames, per_frame, seed=int(threaded))
    cap_map = len(map_pts) + 100
    rng = np.random.default_rng(4)
    ids_all = np.stack([local_ids(f, owner, n_points, cap_map, rng) for f in range(n_frames)])
    ex = orbfe.ORBextractor(*ARGS, device=0, max_batch=slot)
    mp = orbfe.MapPoints(ex, cap_map)
    mp.update(np.arange(len(map_pts)), map_pts.view(orbfe.WP_DTYPE), map_desc)
    st = ex.stream(slots=3, slot_frames=slot)
    st.enable_track(mp, n_points, GRID[0], GRID[1], 0.0, 0.0, float(W), float(H))
    got = run_ring(st, frames, frusta_p, ids_all, slot, 20.0, 0.85, threaded)
    assert len(got) == n_frames
    trk = orbfe.FrameTracker(ex, GRID[0], GRID[1], 0.0, 0.0, float(W), float(H))
    total = 0
    for f in range(n_frames):
        kp_r, desc_r, per_r, match_r, n_r, pts_f, mpd_f = oracle_frame(eo, frames[f], frusta_o[f], ids_all[f], map_pts, map_desc, 20.0, 0.85)
        kp, desc, per, match, nm = got[f]
        assert kp.tobytes() == kp_r.tobytes() and np.array_equal(desc, desc_r) and np.array_equal(per, per_r), f
        assert nm == n_r and np.array_equal(match, match_r), "frame %d: %d vs %d matches" % (f, nm, n_r)
        total += n_r
        if f % 9 == 0:  # the single-frame entry points on the same inputs: explicit points, and ids into the resident map
            one = trk.TrackFrame(frames[f], frusta_p[f], pts_f.view(orbfe.WP_DTYPE), mpd_f, 20.0, 0.85)
            assert one["nmatches"] == nm and np.array_equal(one["match"], match)
            byid = trk.TrackFrameMap(frames[f], frusta_p[f], mp, ids_all[f], 20.0, 0.85)
            assert byid["nmatches"] == nm and np.array_equal(byid["match"], match) and byid["kp"].tobytes() == kp_r.tobytes()
            # (a record whose id is skipped / outside the map carries no point data; everything else is the explicit call's)
            real = (ids_all[f] >= 0) & (ids_all[f] < cap_map)
            assert byid["mps"][real].tobytes() == one["mps"][real].tobytes() and byid["proj_xr"][real].tobytes() == one["proj_xr"][real].tobytes()
            assert (byid["mps"]["in_view"][~real] == 0).all()
    assert total > 80 * n_frames
    st.close()
    mp.close()


def test_map_update_ordering_and_mixed_submissions(built):
    """orbfe_map_update takes effect for the submissions made after it; plain submissions and extract-and-match ones share
    the ring; collect_track refuses a plain submission; collect works on either."""
    import orbfe
    W, H = ARGS[6], ARGS[7]
    slot = 4
    eo, frames, frusta_o, frusta_p, map_pts, map_desc, owner = build_scene(slot, 300, seed=7)
    ex = orbfe.ORBextractor(*ARGS, device=0, max_batch=slot)
    cap_map = len(map_pts)
    mp = orbfe.MapPoints(ex, cap_map)
    st = ex.stream(slots=2, slot_frames=slot)
    with pytest.raises(orbfe.OrbfeError):  # not enabled yet
        st._grid = (GRID[0], GRID[1], 0.0, 0.0, 0.1, 0.1)
        st.submit_track(frames, frusta_p, np.zeros((slot, 10), np.int32), 20.0, 0.85)
    st.enable_track(mp, 400, GRID[0], GRID[1], 0.0, 0.0, float(W), float(H))
    ids_all = np.stack([np.flatnonzero(owner == f).astype(np.int32) for f in range(slot)])
    # nothing uploaded yet: every entry is "bad" -> no matches
    assert st.submit_track(frames, frusta_p, ids_all, 20.0, 0.85)
    assert st.submit(frames)  # a plain submission behind it
    first = st.collect_track()
    assert all(r[4] == 0 and (r[3] == -1).all() for r in first)
    with pytest.raises(orbfe.OrbfeError):
        st.collect_track()  # the oldest submission is the plain one
    plain = st.collect()
    for a, b in zip(first, plain):
        assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1])
    mp.update(np.arange(cap_map), map_pts.view(orbfe.WP_DTYPE), map_desc)
    assert st.submit_track(frames, frusta_p, ids_all, 20.0, 0.85)
    second = st.collect_track()
    for f in range(slot):
        _, _, _, match_r, n_r, _, _ = oracle_frame(eo, frames[f], frusta_o[f], ids_all[f], map_pts, map_desc, 20.0, 0.85)
        assert second[f][4] == n_r and np.array_equal(second[f][3], match_r) and n_r > 80
    # a changed observation count / bad flag is seen by the next submission
    changed = map_pts.copy()
    changed["bad"][::2] = 1
    mp.update(np.arange(0, cap_map, 2), changed[::2].view(orbfe.WP_DTYPE), map_desc[::2])
    assert st.submit_track(frames, frusta_p, ids_all, 20.0, 0.85)
    third = st.collect_track()
    for f in range(slot):
        _, _, _, match_r, n_r, _, _ = oracle_frame(eo, frames[f], frusta_o[f], ids_all[f], changed, map_desc, 20.0, 0.85)
        assert third[f][4] == n_r and np.array_equal(third[f][3], match_r) and n_r < second[f][4]
    with pytest.raises(orbfe.OrbfeError):
        mp.update([cap_map], map_pts[:1].view(orbfe.WP_DTYPE), map_desc[:1])  # id outside the map
    st.close()
    mp.close()
