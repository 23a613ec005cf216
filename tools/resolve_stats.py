#!/usr/bin/env python3
"""Sweeps and exact rescans of proj_resolve_kernel on frames of the bench stream (host entry point; liborbfe_diag.so + ORBFE_DEBUG_MATCH)."""
import os
import sys

os.environ["ORBFE_DEBUG_MATCH"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import orbfe  # noqa: E402
orbfe.LIB_PATH = os.path.join(orbfe.CSRC, "liborbfe_diag.so")  # the counters exist only in the -DORBFE_DIAG build (make diag)
from orbfe import synth  # noqa: E402

ARGS = (1000, 40000, 1.2, 8, 20, 7, 752, 480)
ex = orbfe.ORBextractor(*ARGS, device=0, max_batch=8)
m = orbfe.ORBmatcher(ex)
frames = list(synth.stream(752, 480, 8))
res = ex.extract_batch(frames)
rng = np.random.default_rng(0)
for kp, desc, _ in res:
    kpv = kp.view(orbfe.KP_DTYPE)
    mps, mpd = bench.make_map_points(kpv, len(kpv), desc, 2000, rng, ex.nlevels, orbfe.MP_DTYPE)
    fv = orbfe.make_frame_view(kpv, desc, 64, 48, 0.0, 0.0, 752.0, 480.0, ex.mvScaleFactor)
    n, out = m.SearchByProjection(fv, mps, mpd, 20.0, False, 0.0, 0.85, None)
    print("matches", n, flush=True)
