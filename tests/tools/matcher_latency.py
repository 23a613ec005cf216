#!/usr/bin/env python3
"""Per-call wall time of every matcher entry point of the C ABI (host pointers in and out: pinned staging, one
H2D, the kernels, one D2H, stream sync) next to the single-thread C oracle on the same inputs, at the sizes of
BASELINE config 1 (752x480, ~1000 keypoints).  Results are checked equal before timing.  DESIGN.md section 6
quotes this table; it is never the bench's `value`.

usage: python3 tests/tools/matcher_latency.py [--reps 200] [--json out.json]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import frustum_scenarios as FS  # noqa: E402
import match_scenarios as S  # noqa: E402
import oracle_py as O  # noqa: E402
import orbfe  # noqa: E402
import test_distinct  # noqa: E402
import test_fuse  # noqa: E402
import test_sim3_reloc as T3  # noqa: E402
import test_triangulation  # noqa: E402
import vocab_synth as vs  # noqa: E402
from orbfe import synth  # noqa: E402
from test_frustum import ON, OP, PN  # noqa: E402

W, H = 752, 480
ARGS = (1000, 40000, 1.2, 8, 20, 7, W, H)
MP_NAMES_O = ("projX", "projY", "viewCos", "trackDepth", "level", "inView", "bad", "observations")


def timed(fn, reps):
    fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--cpu-reps", type=int, default=5)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    eo = O.Extractor(*ARGS)
    ex = orbfe.ORBextractor(*ARGS)
    m = orbfe.ORBmatcher(ex)
    frames = list(synth.stream(W, H, 2))
    kp, desc, _ = eo.extract(frames[0])
    kpb, descb, _ = eo.extract(frames[1])
    n = len(kp)
    kpp, kpbp = kp.view(orbfe.KP_DTYPE), kpb.view(orbfe.KP_DTYPE)
    sf = eo.scaleFactors
    fvo = O.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, float(W), float(H), sf)
    fv = orbfe.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
    fvbo = O.make_frame_view(kpb, descb, 64, 48, 0.0, 0.0, float(W), float(H), sf)
    fvb = orbfe.make_frame_view(kpb, descb, 64, 48, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
    rows = []

    def row(name, ref, size, gpu_fn, cpu_fn, same):
        g, c = gpu_fn(), cpu_fn()
        assert same(g, c), name
        tg, tc = timed(gpu_fn, a.reps), timed(cpu_fn, a.cpu_reps)
        rows.append(dict(entry=name, reference=ref, size=size, gpu_ms_per_call=tg, oracle_ms_per_call=tc, ratio=tc / tg))
        print("%-36s %-44s gpu %7.3f ms   oracle(1 thread) %8.3f ms   x%.1f" % (name, size, tg, tc, tc / tg), flush=True)

    eq2 = lambda g, c: g[0] == c[0] and np.array_equal(g[1], c[1])
    eqp = lambda g, c: np.array_equal(g[0], c[0]) and np.array_equal(g[1], c[1])

    # a13
    M = 2000
    mps, mpd, init_obs = S.projection_scenario(kp, desc, M, 1, O.MP_DTYPE, MP_NAMES_O, 8)
    row("orbfe_match_projection", "ORBmatcher.cc:31-123", "N=%d M=%d th=20" % (n, M),
        lambda: m.SearchByProjection(fv, mps.view(orbfe.MP_DTYPE), mpd, 20.0, False, 0.0, 0.85, init_obs),
        lambda: O.search_by_projection(fvo, mps, mpd, init_obs, 20.0, 0.85), eq2)
    # a14
    kf_off, kf_idx, f_off, f_idx, has = S.bow_scenario(kp, desc, kpb, descb, 100, 3)
    row("orbfe_match_bow", "ORBmatcher.cc:133-327", "N=%d/%d nodes=%d" % (n, len(kpb), len(kf_off) - 1),
        lambda: m.SearchByBoW(kf_off, kf_idx, f_off, f_idx, desc, kp["angle"], has, descb, kpb["angle"], 0.75, True),
        lambda: O.search_by_bow(kf_off, kf_idx, f_off, f_idx, desc, kp["angle"], has, descb, kpb["angle"], 0.75, True), eq2)
    # f1
    row("orbfe_match_initialization", "ORBmatcher.cc:329-439", "N=%d/%d window=100" % (n, len(kpb)),
        lambda: m.SearchForInitialization(fv, fvb, 100, 0.9, True),
        lambda: O.search_for_initialization(fvo, fvbo, 100, 0.9, True), eq2)
    # f3
    Fo, Fp = O.Frustum(), orbfe.Frustum()
    v = FS.fill_frustum(Fo, ON, seed=3)
    FS.fill_frustum(Fp, PN, seed=3)
    cloud = FS.world_points(4000, O.WP_DTYPE, OP, seed=5)
    row("orbfe_project_map_points", "Frame.cc:272-331", "M=4000",
        lambda: m.isInFrustum_batch(Fp, cloud.view(orbfe.WP_DTYPE)),
        lambda: O.is_in_frustum(Fo, cloud),
        lambda g, c: g[0].tobytes() == c[0].tobytes() and np.array_equal(g[1], c[1]))
    # f2
    pts, fmpd, _, inv_s2 = test_fuse.scenario(kp, desc, sf, v, M, 1, False)
    row("orbfe_fuse_search", "ORBmatcher.cc:678-836", "N=%d M=%d th=3" % (n, M),
        lambda: m.Fuse_search(fv, inv_s2, None, Fp, 3.0, pts.view(orbfe.WP_DTYPE), fmpd),
        lambda: O.fuse_search(fvo, inv_s2, None, Fo, 3.0, pts, fmpd), eqp)
    # the same search against a RESIDENT key frame with the map points named by id out of a resident map (SearchInNeighbors' shape)
    kf_res_f = orbfe.KeyFrame(ex, kpp, desc, np.full(n, -1, np.int32), ex.mvScaleFactor)
    kf_res_f.set_grid(64, 48, 0.0, 0.0, float(W), float(H), inv_s2, None)
    mp_res_f = orbfe.MapPoints(ex, M)
    st_f = pts.copy()
    st_f["skip"] = 0
    mp_res_f.update(np.arange(M), st_f.view(orbfe.WP_DTYPE), fmpd)
    ids_f = np.where(pts["skip"] != 0, ~np.arange(M, dtype=np.int32), np.arange(M, dtype=np.int32)).astype(np.int32)
    row("orbfe_fuse_search_keyframe", "LocalMapping.cc:764-860", "N=%d M=%d th=3, resident key frame + map" % (n, M),
        lambda: m.Fuse_search_keyframe(kf_res_f, mp_res_f, ids_f, Fp, 3.0),
        lambda: O.fuse_search(fvo, inv_s2, None, Fo, 3.0, pts, fmpd), eqp)
    row("orbfe_fuse_search_sim3", "ORBmatcher.cc:864-975", "N=%d M=%d th=4" % (n, M),
        lambda: m.Fuse_search_sim3(fv, Fp, 4.0, pts.view(orbfe.WP_DTYPE), fmpd),
        lambda: O.fuse_search_sim3(fvo, Fo, 4.0, pts, fmpd), eqp)
    off1, idx1, off2, idx2, kp2, d2, h1, h2, s1, s2, F12, ep = test_triangulation.scenario(kp, desc, 2, True, False)
    off1, idx1, off2, idx2 = (np.asarray(x, np.int32) for x in (off1, idx1, off2, idx2))  # not the Python list conversion
    h1, h2 = h1.astype(np.uint8), h2.astype(np.uint8)
    row("orbfe_match_triangulation", "ORBmatcher.cc:441-676", "N=%d/%d nodes=%d" % (n, len(kp2), len(off1) - 1),
        lambda: m.SearchForTriangulation(off1, idx1, off2, idx2, kpp, desc, h1, s1, kp2.view(orbfe.KP_DTYPE), d2, h2, s2,
                                         ex.mvScaleFactor, F12, ep, False, False, True),
        lambda: O.search_for_triangulation(off1, idx1, off2, idx2, kp, desc, h1, s1, kp2, d2, h2, s2, sf, F12, ep, False,
                                           False, True), eq2)
    # the mapping thread's call shape (src/LocalMapping.cc:455-488): ONE key frame against K = 20 neighbours
    import test_triangulation_batch as TB
    K = 20
    node1 = TB.nodes_of(kp)
    nbs = [TB.neighbour(kp, desc, 500 + k, True, False) for k in range(K)]
    csrs = [tuple(np.asarray(x, np.int32) for x in TB.csr(node1, nb["node"])) for nb in nbs]
    kf1 = orbfe.KeyFrame(ex, kpp, desc, node1, ex.mvScaleFactor)
    kf2 = [orbfe.KeyFrame(ex, nb["kp"].view(orbfe.KP_DTYPE), nb["desc"], nb["node"], ex.mvScaleFactor) for nb in nbs]
    prm = [orbfe.tri_params(nb["F12"], nb["ep"], False, False, True) for nb in nbs]
    has2 = [nb["has"] for nb in nbs]

    import ctypes as C
    kfp = (C.c_void_p * K)(*[k.h.value for k in kf2])
    h2p = (C.c_void_p * K)(*[v.ctypes.data for v in has2])
    P = (orbfe.TriParams * K)(*prm)
    raw = np.full((K, kf1.n), -1, np.int32)
    rbin = np.zeros((K, kf1.n), np.uint8)
    vp = lambda a_: a_.ctypes.data_as(C.c_void_p)

    def batch_gpu():
        # the C call with pre-built argument arrays (what a C++ caller pays) ...
        rc = ex.L.orbfe_match_triangulation_batch(ex.h, kf1.h, vp(h1), K, kfp, h2p, P, vp(raw), vp(rbin))
        assert rc == 0
        return raw, rbin

    def replay(res):
        # ... and the host replay, one orbfe_triangulation_select per neighbour (1.5 us each in C: tests/cpp/test_adaptor.cpp
        # prints tri_select_us; through ctypes it is ~10 us per call, so it is timed apart from the GPU call)
        return [orbfe.triangulation_select(res[0][k], res[1][k], h1, True) for k in range(K)]

    def seq_gpu():
        return [m.SearchForTriangulation(*csrs[k], kpp, desc, h1, None, nbs[k]["kp"].view(orbfe.KP_DTYPE), nbs[k]["desc"], has2[k],
                                         None, ex.mvScaleFactor, nbs[k]["F12"], nbs[k]["ep"], False, False, True) for k in range(K)]

    def seq_cpu():
        return [O.search_for_triangulation(*csrs[k], kp, desc, h1, None, nbs[k]["kp"], nbs[k]["desc"], has2[k], None, sf,
                                           nbs[k]["F12"], nbs[k]["ep"], False, False, True) for k in range(K)]

    eqk = lambda g, c: all(a_[0] == b_[0] and np.array_equal(a_[1], b_[1]) for a_, b_ in zip(g, c))
    row("orbfe_match_triangulation_batch K=20", "LocalMapping.cc:455-488", "N=%d, 20 resident neighbours, C call" % n, batch_gpu, seq_cpu,
        lambda g, c: eqk(replay(g), c))
    row("orbfe_match_triangulation x 20", "LocalMapping.cc:455-488", "N=%d, 20 single calls" % n, seq_gpu, seq_cpu, eqk)
    # SearchBySim3
    sc = T3.sim3_scenario(kp, desc, sf, 1)
    d12o, d21o, d12, d21 = O.Sim3Dir(), O.Sim3Dir(), orbfe.Sim3View(), orbfe.Sim3View()
    sc["fill12"](d12o, T3.SN_O), sc["fill21"](d21o, T3.SN_O), sc["fill12"](d12, T3.SN_P), sc["fill21"](d21, T3.SN_P)
    fv2o = O.make_frame_view(sc["kp2"], sc["desc2"], 64, 48, 0.0, 0.0, float(W), float(H), sf)
    fv2 = orbfe.make_frame_view(sc["kp2"], sc["desc2"], 64, 48, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
    row("orbfe_search_by_sim3", "ORBmatcher.cc:977-1200", "N=%d/%d th=7.5" % (n, len(sc["kp2"])),
        lambda: m.SearchBySim3(fv, fv2, d12, d21, sc["mp1"].view(orbfe.WP_DTYPE), sc["mpd1"], sc["mp2"].view(orbfe.WP_DTYPE),
                               sc["mpd2"], 7.5),
        lambda: O.search_by_sim3(fvo, fv2o, d12o, d21o, sc["mp1"], sc["mpd1"], sc["mp2"], sc["mpd2"], 7.5), eq2)
    # relocalisation
    rpts, rmpd, rang, rhas = T3.reloc_scenario(kp, desc, sf, v, M, 2)
    row("orbfe_match_projection_keyframe", "ORBmatcher.cc:1202-1326", "N=%d M=%d th=12" % (n, M),
        lambda: m.SearchByProjection_keyframe(fv, Fp, rpts.view(orbfe.WP_DTYPE), rmpd, rang, rhas, 12.0, True),
        lambda: O.search_by_projection_kf(fvo, Fo, rpts, rmpd, rang, rhas, 12.0, True), eq2)
    # distinct
    doff, ddesc = test_distinct.make_sets(3, [int(x) for x in np.random.default_rng(0).integers(2, 30, 500)])
    row("orbfe_distinctive_descriptors", "MapPoint.cc:343-416", "500 map points, %d observations" % len(ddesc),
        lambda: m.ComputeDistinctiveDescriptors(doff, ddesc),
        lambda: O.distinctive_descriptors(doff, ddesc), eqp)
    # f4
    t = vs.make_tree(10, 6, seed=16, early_leaf_p=0.02)
    feats = vs.features_near(t, n, seed=n)
    voc = orbfe.ORBVocabulary(ex, t["childOff"], t["childIdx"], t["nodeDesc"], t["wordId"], t["weight"], 6)
    row("orbfe_bow_transform", "TemplatedVocabulary.h:1227-1270", "n=%d k=10 L=6 (%d nodes)" % (n, len(t["wordId"])),
        lambda: voc.transform(feats, 4),
        lambda: O.vocab_transform(t["childOff"], t["childIdx"], t["nodeDesc"], t["wordId"], t["weight"], 6, feats, 4),
        lambda g, c: np.array_equal(g[0], c[0]) and np.array_equal(g[1], c[1]))
    # the chain of a frame tracked against its reference key frame, as one submission and as the three calls it replaces
    ti = vs.spread_first_level(vs.make_tree(10, 6, seed=17, early_leaf_p=0.02), 18)
    voci = orbfe.ORBVocabulary(ex, ti["childOff"], ti["childIdx"], ti["nodeDesc"], ti["wordId"], ti["weight"], 6)
    tv = lambda d: O.vocab_transform(ti["childOff"], ti["childIdx"], ti["nodeDesc"], ti["wordId"], ti["weight"], 6, d, 4)
    # mFeatVec holds a feature only when its word's weight is > 0 (TemplatedVocabulary.h:1168-1172): node -1 otherwise
    (_, nk_, wk_), (_, nf_, wf_) = tv(desc), tv(descb)
    node_kf, node_f = np.where(wk_ > 0, nk_, -1).astype(np.int32), np.where(wf_ > 0, nf_, -1).astype(np.int32)
    res_kf = orbfe.KeyFrame(ex, kpp, desc, node_kf, sf)
    has_kf = (np.random.default_rng(4).random(n) < 0.8).astype(np.uint8)
    shared = sorted((set(node_kf.tolist()) & set(node_f.tolist())) - {-1})
    ko, ki, fo, fi = [0], [], [0], []
    for g in shared:
        ki += list(np.flatnonzero(node_kf == g)); fi += list(np.flatnonzero(node_f == g))
        ko.append(len(ki)); fo.append(len(fi))
    trk = orbfe.FrameTracker(ex, 64, 48, 0.0, 0.0, float(W), float(H))
    import torch
    pinned = torch.from_numpy(frames[1].copy()).pin_memory().numpy()

    def ref_three_calls():
        k2, d2 = ex.extractFeatures(pinned)
        voci.transform(d2, 4)
        return m.SearchByBoW(ko, ki, fo, fi, desc, kp["angle"], has_kf, d2, k2["angle"], 0.75, True)

    def ref_oracle():
        k2, d2, _ = eo.extract(frames[1])
        tv(d2)
        return O.search_by_bow(ko, ki, fo, fi, desc, kp["angle"], has_kf, d2, k2["angle"], 0.75, True)

    def ref_fused():
        r = trk.TrackReferenceKeyFrame(pinned, voci, 4, res_kf, has_kf, 0.75, True)
        return r["nmatches"], r["match"]

    row("orbfe_track_reference_keyframe", "Tracking.cc:825-835", "N=%d/%d, %d shared nodes, one submission" % (n, len(kpb), len(shared)),
        ref_fused, ref_oracle, eq2)
    row("extract + bow_transform + match_bow", "Tracking.cc:825-835", "the same chain as three calls (CSR prebuilt)", ref_three_calls,
        ref_oracle, eq2)
    # the chain of a frame during monocular initialisation (Tracking.cc:566-607): one submission against a resident initial
    # frame, and the two calls it replaces (the reference's own parameters: window 40, ratio 0.45, orientation on)
    ini = orbfe.InitialFrame(ex, kpp, desc)

    def ini_two_calls():
        k2, d2 = ex.extractFeatures(pinned)
        f2 = orbfe.make_frame_view(k2, d2, 64, 48, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
        return m.SearchForInitialization(fv, f2, 40, 0.45, True)

    def ini_oracle():
        k2, d2, _ = eo.extract(frames[1])
        f2 = O.make_frame_view(k2, d2, 64, 48, 0.0, 0.0, float(W), float(H), sf)
        return O.search_for_initialization(fvo, f2, 40, 0.45, True)

    def ini_fused():
        r = trk.TrackInitialization(pinned, ini, 40, 0.45, True)
        return r["nmatches"], r["matches12"]

    row("orbfe_track_initialization", "Tracking.cc:566-607", "N=%d/%d window=40, one submission" % (n, len(kpb)), ini_fused, ini_oracle, eq2)
    row("extract + match_initialization", "Tracking.cc:566-607", "the same chain as two calls", ini_two_calls, ini_oracle, eq2)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(dict(host_cpus=os.cpu_count(), reps=a.reps, rows=rows), f, indent=1)


if __name__ == "__main__":
    main()
