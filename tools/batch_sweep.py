#!/usr/bin/env python3
"""Stage times of the extraction chain per batch size (frames resident in HBM): where the one-launch pyramid kernel
(one block per frame) overtakes the per-level tile launches.  Usage: batch_sweep.py [batch ...]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python"))
import numpy as np, torch
import orbfe
from orbfe import synth
cfg = (1000, 40000, 1.2, 8, 20, 7, 752, 480)
W, H = cfg[6], cfg[7]
batches = [int(a) for a in sys.argv[1:]] or [1, 4, 16, 32, 64, 128, 256]
frames = np.stack(list(synth.stream(W, H, max(batches))))
dev = torch.device("cuda", 0)
d_gray = torch.from_numpy(frames).to(dev)
for B in batches:
    ex = orbfe.ORBextractor(*cfg, device=0, max_batch=B)
    cap = ex.cap
    kp = torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev); desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    n = torch.zeros(B, dtype=torch.int32, device=dev); per = torch.zeros((B, ex.nlevels), dtype=torch.int32, device=dev)
    run = lambda: ex.extract_batch_device(d_gray.data_ptr(), W * H, W, B, kp.data_ptr(), desc.data_ptr(), n.data_ptr(), per.data_ptr(), 0)
    for _ in range(3): run()
    torch.cuda.synchronize()
    ex.set_stage_timing(True)
    for _ in range(20): run()
    torch.cuda.synchronize()
    ms, calls = ex.stage_ms()
    print(json.dumps({"batch": B, **{k: round(v / calls, 4) for k, v in ms.items()}}))
    del ex
