#!/usr/bin/env python3
"""Per-kernel GPU time of the single-frame chains, for `rocprofv3 --kernel-trace --stats -- python3 tools/chain_kernel_times.py <chain>`:
runs 200 calls of one chain (track_initialization | fuse_keyframe | track_frame) on the C1 geometry so that the stats table shows
what each kernel of that chain costs per call."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import orbfe  # noqa: E402
from orbfe import synth  # noqa: E402

W, H = 752, 480
ARGS = (1000, 40000, 1.2, 8, 20, 7, W, H)
chain = sys.argv[1] if len(sys.argv) > 1 else "track_initialization"
ex = orbfe.ORBextractor(*ARGS)
trk = orbfe.FrameTracker(ex, 64, 48, 0.0, 0.0, float(W), float(H))
frames = list(synth.stream(W, H, 2))
pinned = torch.from_numpy(frames[1].copy()).pin_memory().numpy()
kp, desc = ex.extractFeatures(frames[0])
if chain == "track_initialization":
    ini = orbfe.InitialFrame(ex, kp, desc)
    for _ in range(200):
        r = trk.TrackInitialization(pinned, ini, 40, 0.45, True)
    print("matches", r["nmatches"], "level-0 keypoints of the initial frame", int((kp["octave"] == 0).sum()))
elif chain == "fuse_keyframe":
    import frustum_scenarios as FS
    import oracle_py as O
    import test_fuse
    from test_frustum import ON, PN
    Fo, Fp = O.Frustum(), orbfe.Frustum()
    v = FS.fill_frustum(Fo, ON, seed=3)
    FS.fill_frustum(Fp, PN, seed=3)
    pts, mpd, _, inv_s2 = test_fuse.scenario(kp.view(O.KP_DTYPE), desc, ex.mvScaleFactor, v, 2000, 1, False)
    kf = orbfe.KeyFrame(ex, kp, desc, np.full(len(kp), -1, np.int32), ex.mvScaleFactor)
    kf.set_grid(64, 48, 0.0, 0.0, float(W), float(H), inv_s2, None)
    mp = orbfe.MapPoints(ex, 2000)
    st = pts.copy()
    st["skip"] = 0
    mp.update(np.arange(2000), st.view(orbfe.WP_DTYPE), mpd)
    ids = np.arange(2000, dtype=np.int32)
    m = orbfe.ORBmatcher(ex)
    for _ in range(200):
        bi, bd = m.Fuse_search_keyframe(kf, mp, ids, Fp, 3.0)
    print("fused", int((bd <= 30).sum()))
else:
    raise SystemExit("unknown chain " + chain)
