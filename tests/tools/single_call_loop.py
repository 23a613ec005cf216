#!/usr/bin/env python3
"""300 single-frame calls of orbfe_match_projection and orbfe_match_initialization at config-1 sizes, for
`rocprofv3 --kernel-trace --stats -- python3 tests/tools/single_call_loop.py`: per-kernel durations of the live,
one-frame-at-a-time use."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import match_scenarios as S  # noqa: E402
import oracle_py as O  # noqa: E402
import orbfe  # noqa: E402
from orbfe import synth  # noqa: E402

W, H = 752, 480
ARGS = (1000, 40000, 1.2, 8, 20, 7, W, H)
ex = orbfe.ORBextractor(*ARGS)
m = orbfe.ORBmatcher(ex)
kp, desc = ex.extractFeatures(synth.frame(W, H, 1))
names = ("projX", "projY", "viewCos", "trackDepth", "level", "inView", "bad", "observations")
mps, mpd, io = S.projection_scenario(kp.view(O.KP_DTYPE), desc, 2000, 1, O.MP_DTYPE, names, 8)
fv = orbfe.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
for _ in range(300):
    m.SearchByProjection(fv, mps.view(orbfe.MP_DTYPE), mpd, 20.0, False, 0.0, 0.85, io)
f2 = list(synth.stream(W, H, 2))
k1, d1 = ex.extractFeatures(f2[0])
k2, d2 = ex.extractFeatures(f2[1])
g1 = orbfe.make_frame_view(k1, d1, 64, 48, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
g2 = orbfe.make_frame_view(k2, d2, 64, 48, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
for _ in range(300):
    m.SearchForInitialization(g1, g2, 100, 0.9, True)
import test_triangulation  # noqa: E402
import numpy as np  # noqa: E402
kf_off, kf_idx, f_off, f_idx, has = S.bow_scenario(k1, d1, k2, d2, 100, 3)
for _ in range(300):
    m.SearchByBoW(kf_off, kf_idx, f_off, f_idx, d1, k1["angle"], has, d2, k2["angle"], 0.75, True)
off1, idx1, off2, idx2, kpt2, dt2, h1, h2, s1_, s2_, F12, ep = test_triangulation.scenario(k1.view(O.KP_DTYPE), d1, 2, True, False)
off1, idx1, off2, idx2 = (np.asarray(x, np.int32) for x in (off1, idx1, off2, idx2))
h1, h2 = h1.astype(np.uint8), h2.astype(np.uint8)
for _ in range(300):
    m.SearchForTriangulation(off1, idx1, off2, idx2, k1, d1, h1, None, kpt2.view(orbfe.KP_DTYPE), dt2, h2, None, ex.mvScaleFactor, F12, ep,
                             False, False, True)
