#!/usr/bin/env python3
"""20 single-frame SearchByProjection calls on the bench scenario (for a rocprofv3 kernel trace of the host entry point)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import orbfe  # noqa: E402
from orbfe import synth  # noqa: E402

ARGS = (1000, 40000, 1.2, 8, 20, 7, 752, 480)
ex = orbfe.ORBextractor(*ARGS, device=0, max_batch=1)
m = orbfe.ORBmatcher(ex)
kp, desc = ex.extractFeatures(next(iter(synth.stream(752, 480, 1))))
rng = np.random.default_rng(1)
mps, mpd = bench.make_map_points(kp, len(kp), desc, 2000, rng, ex.nlevels, orbfe.MP_DTYPE)
fv = orbfe.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, 752.0, 480.0, ex.mvScaleFactor)
for _ in range(3):
    m.SearchByProjection(fv, mps, mpd, 20.0, False, 0.0, 0.85, None)
t = time.perf_counter()
for _ in range(20):
    n, _o = m.SearchByProjection(fv, mps, mpd, 20.0, False, 0.0, 0.85, None)
print("SearchByProjection %.3f ms/call, %d matches" % ((time.perf_counter() - t) / 20 * 1e3, n))
