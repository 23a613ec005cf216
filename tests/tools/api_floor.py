"""Latency floor of one host-pointer call (pinned staging, H2D, one kernel, D2H, stream sync) at vanishing problem size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, ctypes as C
import orbfe, frustum_scenarios as FS
from test_frustum import PN, PP
ex = orbfe.ORBextractor(1000, 40000, 1.2, 8, 20, 7, 752, 480); m = orbfe.ORBmatcher(ex)
F = orbfe.Frustum(); FS.fill_frustum(F, PN, seed=3)
def timed(fn, reps=500):
    fn(); t = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t) / reps * 1e6
for M in (1, 100, 4000):
    cloud = FS.world_points(M, orbfe.WP_DTYPE, PP, seed=5)
    print("project_map_points M=%d: %.1f us" % (M, timed(lambda: m.isInFrustum_batch(F, cloud))))
# raw ctypes call cost without numpy wrappers
L = ex.L
cloud = FS.world_points(1, orbfe.WP_DTYPE, PP, seed=5)
out = np.zeros(1, orbfe.MP_DTYPE); xr = np.zeros(1, np.float32)
f = lambda: L.orbfe_project_map_points(ex.h, C.byref(F), 1, cloud.ctypes.data, out.ctypes.data, xr.ctypes.data)
print("raw ctypes M=1: %.1f us" % timed(f))
print("orbfe_hamming (no GPU): %.1f us" % timed(lambda: L.orbfe_hamming(cloud.ctypes.data, cloud.ctypes.data)))
