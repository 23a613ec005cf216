#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host-pointer batched API (orbfe_extract_batch: host memcpy into pinned staging,
H2D, kernels, D2H, sync) for DESIGN.md section 6 -- never the bench's `value`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np
import orbfe, bench
from orbfe import synth
cfg = bench.WORKLOADS["euroc_752x480"]
for B in (1, 16, 64, 256):
    ex = orbfe.ORBextractor(*cfg, device=0, max_batch=B)
    frames = list(synth.stream(cfg[6], cfg[7], B))
    for _ in range(3):
        ex.extract_batch(frames)
    n = max(3, 512 // B)
    t = time.perf_counter()
    for _ in range(n):
        ex.extract_batch(frames)
    dt = (time.perf_counter() - t) / n
    print("batch %3d: %.3f ms per call, %.0f frames/s (host pointers in and out)" % (B, dt * 1e3, B / dt))
