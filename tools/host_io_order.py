import sys, time
sys.path.insert(0, "orb_slam3_v1.0_amd/python"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import bench, orbfe
from orbfe import synth
cfg = bench.WORKLOADS["euroc_752x480"]
ex = orbfe.ORBextractor(*cfg, device=0, max_batch=512)
frames = np.stack(list(synth.stream(752, 480, 512)))
for name, pinned in (("pinned", True), ("pageable", False), ("pinned", True), ("pageable", False), ("pinned", True)):
    t = time.time(); r = bench.host_io_rate(ex, frames, 256, 40, pinned); print("%-9s %8.0f frames/s (%.1f s)" % (name, r, time.time() - t), flush=True)
r, m = bench.host_io_match_rate(ex, frames, 256, 20); print("match     %8.0f frames/s" % r)
r = bench.host_io_rate(ex, frames, 256, 40, True); print("pinned    %8.0f frames/s" % r)
