#!/usr/bin/env python3
"""The fused per-frame chains (orbfe_track_frame; with `ref` as first argument orbfe_track_reference_keyframe) in a loop, for
`rocprofv3 --kernel-trace --stats`: per-kernel durations at batch 1 and -- with `--report DIR [last-kernel]` on the trace's
kernel CSV -- the timeline of one call (kernel start / end relative to the call's first kernel, gaps between kernels)."""
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def report(d, last="proj_resolve"):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # one call = the dispatches from one memset-following pyramid kernel to the resolve kernel; take the LAST complete call
    ends = [i for i, r in enumerate(rows) if last in r["Kernel_Name"]]
    if len(ends) < 2:
        raise SystemExit("no complete call in the trace")
    lo, hi = ends[-2] + 1, ends[-1]
    t0 = int(rows[lo]["Start_Timestamp"])
    prev_end = t0
    print("%-44s %9s %9s %8s %7s" % ("kernel", "start us", "dur us", "gap us", "grid"))
    for r in rows[lo:hi + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void orbfe::", "").replace("(anonymous namespace)::", "")[:44]
        print("%-44s %9.1f %9.1f %8.1f %7s" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size", "")))
        prev_end = e
    print("GPU span of the call: %.1f us; sum of kernel durations %.1f us" % (
        (prev_end - t0) / 1e3, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[lo:hi + 1]) / 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        report(sys.argv[2], *sys.argv[3:4])
        sys.exit(0)
    import numpy as np
    import torch
    import bench
    import frustum_scenarios as FS
    import orbfe
    from orbfe import synth
    cfg = bench.WORKLOADS["euroc_752x480"]
    W, H = cfg[6], cfg[7]
    ex = orbfe.ORBextractor(*cfg, device=0, max_batch=1)
    trk = orbfe.FrameTracker(ex, bench.GRID[0], bench.GRID[1], 0.0, 0.0, float(W), float(H))
    frames = [torch.from_numpy(f.copy()).pin_memory().numpy() for f in synth.stream(W, H, 16)]
    kp, desc = ex.extractFeatures(frames[0])
    if len(sys.argv) > 1 and sys.argv[1] == "ref":  # the chain against the reference key frame (frame 0), vocabulary k=10 L=5
        import vocab_synth as vs
        t = vs.spread_first_level(vs.make_tree(10, 5, seed=17, early_leaf_p=0.02), 18)
        voc = orbfe.ORBVocabulary(ex, t["childOff"], t["childIdx"], t["nodeDesc"], t["wordId"], t["weight"], 5)
        _, node, _ = voc.transform(desc, 3)
        kf = orbfe.KeyFrame(ex, kp, desc, node, ex.mvScaleFactor)
        has = np.ones(len(kp), np.uint8)
        for i in range(10):
            trk.TrackReferenceKeyFrame(frames[i % 16], voc, 3, kf, has)
        ts = []
        for i in range(100):
            t0 = time.perf_counter()
            r = trk.TrackReferenceKeyFrame(frames[i % 16], voc, 3, kf, has)
            ts.append(time.perf_counter() - t0)
        print("orbfe_track_reference_keyframe: median %.1f us, min %.1f us per call, %d nodes in the key frame, %d keypoints, %d matches" % (
            np.median(ts) * 1e6, min(ts) * 1e6, len(set(node.tolist())), len(r["kp"]), r["nmatches"]))
        sys.exit(0)
    M = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_MAP_POINTS
    Fp = orbfe.Frustum()
    names = {k: k for k in ("rcw", "tcw", "twc", "min_x", "max_x", "min_y", "max_y", "fx", "fy", "cx", "cy", "k1", "k2", "k3", "k4",
                            "mbf", "log_scale_factor", "n_levels", "camera_model")}
    v = FS.fill_frustum(Fp, names, W=float(W), H=float(H), n_levels=ex.nlevels, scale=cfg[2], seed=21)
    pts, mpd = FS.world_points_on_keypoints(kp, desc, v, M, np.random.default_rng(12), ex.nlevels, orbfe.WP_DTYPE)
    for i in range(10):
        trk.TrackFrame(frames[i % 16], Fp, pts, mpd, bench.MATCH_TH, bench.MATCH_NN)
    ts = []
    for i in range(100):
        t = time.perf_counter()
        r = trk.TrackFrame(frames[i % 16], Fp, pts, mpd, bench.MATCH_TH, bench.MATCH_NN)
        ts.append(time.perf_counter() - t)
    print("orbfe_track_frame: median %.1f us, min %.1f us per call, M = %d, %d keypoints, %d matches" % (
        np.median(ts) * 1e6, min(ts) * 1e6, M, len(r["kp"]), r["nmatches"]))
