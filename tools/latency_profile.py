#!/usr/bin/env python3
"""Single-frame host-API loop for rocprofv3 --kernel-trace (per-kernel durations at batch 1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python")); sys.path.insert(0, ROOT)
import orbfe, bench
from orbfe import synth
cfg = bench.WORKLOADS["euroc_752x480"]
ex = orbfe.ORBextractor(*cfg, device=0, max_batch=1)
for f in synth.stream(cfg[6], cfg[7], 60):
    ex.extractFeatures(f)
