#!/usr/bin/env python3
"""Diagnostics: replay the bench's match scenario for a few frames through the host API with
the diagnostics build (make diag) and ORBFE_DEBUG_MATCH=1 (prints sweeps / cooperative rescans), and time the call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python")); sys.path.insert(0, ROOT)
os.environ["ORBFE_DEBUG_MATCH"] = "1"
import numpy as np
import orbfe
orbfe.LIB_PATH = os.path.join(orbfe.CSRC, "liborbfe_diag.so")  # the counters exist only in the -DORBFE_DIAG build (make diag)
from orbfe import synth
import bench
cfg = bench.WORKLOADS["euroc_752x480"]
ex = orbfe.ORBextractor(*cfg, device=0, max_batch=1)
m = orbfe.ORBmatcher(ex)
rng = np.random.default_rng(1234)
for idx, img in enumerate(synth.stream(752, 480, 3)):
    kp, desc = ex.extractFeatures(img)
    mps, mpd = bench.make_map_points(kp, len(kp), desc, 2000, rng, ex.nlevels, orbfe.MP_DTYPE)
    fv = orbfe.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, 752.0, 480.0, ex.mvScaleFactor)
    t = time.perf_counter(); n, out = m.SearchByProjection(fv, mps, mpd, 20.0, False, 0.0, 0.85, None); dt = time.perf_counter() - t
    print("frame", idx, "matches", n, "host-call ms", round(dt * 1e3, 3))
