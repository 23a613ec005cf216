#!/usr/bin/env python3
"""Single-frame latency of the host-pointer API (orbfe_extract: H2D + kernels + D2H + sync) and of
extract+match, for BASELINE.md (config C1 plumbing / C2 stream)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np
import orbfe, bench
from orbfe import synth
if os.environ.get("ORBFE_PROBE_LIB"):   # A/B of two builds
    orbfe.LIB_PATH = os.path.abspath(os.environ["ORBFE_PROBE_LIB"])
for wl in ("euroc_752x480", "batched_1280x720", "tumvi_1024x1024"):
    cfg = bench.WORKLOADS[wl]
    ex = orbfe.ORBextractor(*cfg, device=0, max_batch=1)
    m = orbfe.ORBmatcher(ex)
    frames = list(synth.stream(cfg[6], cfg[7], 40))
    rng = np.random.default_rng(1)
    for f in frames[:5]:
        ex.extractFeatures(f)
    t = time.perf_counter()
    for f in frames:
        kp, desc = ex.extractFeatures(f)
    dt_e = (time.perf_counter() - t) / len(frames)
    # the same frames in pinned host memory (what the reference hands over: cv::cuda::HostMem)
    import torch
    pinned = [torch.from_numpy(f).pin_memory().numpy() for f in frames]
    for f in pinned:  # first DMA access to a fresh pinned allocation maps it: keep that out of the timed loop
        ex.extractFeatures(f)
    t = time.perf_counter()
    for f in pinned:
        ex.extractFeatures(f)
    dt_p = (time.perf_counter() - t) / len(pinned)
    mps, mpd = bench.make_map_points(kp, len(kp), desc, 2000, rng, ex.nlevels, orbfe.MP_DTYPE)
    fv = orbfe.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, float(cfg[6]), float(cfg[7]), ex.mvScaleFactor)
    for _ in range(3):
        m.SearchByProjection(fv, mps, mpd, 20.0, False, 0.0, 0.85, None)
    t = time.perf_counter()
    for _ in range(20):
        n, _o = m.SearchByProjection(fv, mps, mpd, 20.0, False, 0.0, 0.85, None)
    dt_m = (time.perf_counter() - t) / 20
    print("%s: extract %.3f ms/frame pageable (%.0f fps), %.3f ms pinned (%.0f fps), SearchByProjection(2000 MPs) %.3f ms, "
          "%d kp, %d matches" % (wl, dt_e * 1e3, 1 / dt_e, dt_p * 1e3, 1 / dt_p, dt_m * 1e3, len(kp), n))
