"""Per-stage GPU time of a SINGLE-frame extraction (HIP events between the stages, plain launches): what the launch chain of
seven pyramid levels, FAST, quadtree and orient_brief costs when the chip is otherwise empty."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "orb_slam3_v1.0_amd", "python"))
import numpy as np, orbfe
from orbfe import synth
import torch
cfg = (1000, 40000, 1.2, 8, 20, 7, 752, 480)
ex = orbfe.ORBextractor(*cfg, device=0, max_batch=1)
img = torch.from_numpy(synth.frame(752, 480, 3).copy()).pin_memory().numpy()
for _ in range(20): ex.extractFeatures(img)
ex.set_stage_timing(True)
for _ in range(200):
    ex.extractFeatures(img)
st, n = ex.stage_ms()
print({k: round(v * 1000, 1) for k, v in st.items()}, "us per stage (mean of %d calls), plain launches, batch 1" % n)
