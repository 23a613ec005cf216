#!/usr/bin/env python3
"""Experiment: R independent extract+match pipelines (own handle, matcher, streams and buffers each, B frames per step each)
stepping in turn on one GPU -- does the FAST kernel of one pipeline fill the latency-bound phases of the other?
usage: two_pipelines.py [R] [B] [steps]   -> frames/s for R = 1 and R"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import orbfe  # noqa: E402
from orbfe import synth  # noqa: E402


def build(dev, B, sets_gray):
    cfg = bench.WORKLOADS["euroc_752x480"]
    W, H = cfg[6], cfg[7]
    ex = orbfe.ORBextractor(*cfg, device=0, max_batch=B)
    m = orbfe.ORBmatcher(ex)
    cap = ex.cap
    s1 = torch.cuda.Stream(dev)
    state = {}

    def extract_fn(b, fs):
        ex.extract_batch_device(sets_gray[fs].data_ptr(), W * H, W, B, b["kp"].data_ptr(), b["desc"].data_ptr(), b["n"].data_ptr(),
                                b["per"].data_ptr(), r.s1.cuda_stream)

    def match_fn(b, fs):
        m.SearchByProjection_batch_device(B, b["kp"].data_ptr(), b["desc"].data_ptr(), b["n"].data_ptr(), cap, 64, 48, 0.0, 0.0, float(W),
                                          float(H), 2000, state["mps"][fs].data_ptr(), state["mpd"][fs].data_ptr(), None, 20.0, 0.85,
                                          b["match"].data_ptr(), b["nmatch"].data_ptr(), stream=r.s2.cuda_stream)

    with torch.cuda.stream(s1):
        r = bench.StepRunner(dev, B, cap, ex.nlevels, extract_fn, match_fn, None, 1, "none", True, len(sets_gray))
    r.s1 = s1
    mps_d, mpd_d = [], []
    for fs in range(len(sets_gray)):
        extract_fn(r.bufs[0], fs)
        torch.cuda.synchronize(dev)
        b0 = r.bufs[0]
        kp_h = b0["kp"].cpu().numpy().reshape(B, cap * 24).view(orbfe.KP_DTYPE).reshape(B, cap)
        desc_h, n_h = b0["desc"].cpu().numpy(), b0["n"].cpu().numpy()
        rng = np.random.default_rng(fs)
        mps_all, mpd_all = np.zeros((B, 2000), orbfe.MP_DTYPE), np.zeros((B, 2000, 32), np.uint8)
        for i in range(B):
            mps_all[i], mpd_all[i] = bench.make_map_points(kp_h[i], int(n_h[i]), desc_h[i], 2000, rng, ex.nlevels, orbfe.MP_DTYPE)
        mps_d.append(torch.from_numpy(mps_all.view(np.uint8).reshape(-1)).to(dev))
        mpd_d.append(torch.from_numpy(mpd_all.reshape(-1)).to(dev))
    state["mps"], state["mpd"] = mps_d, mpd_d
    return r, ex


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    dev = torch.device("cuda", 0)
    sets = [torch.from_numpy(np.stack(list(synth.stream(752, 480, B, 1000 * s)))).to(dev) for s in range(3)]
    pipes = [build(dev, B, sets) for _ in range(R)]
    for n_active in (1, R):
        act = pipes[:n_active]
        for _ in range(4):
            for r, _e in act:
                r.step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            for r, _e in act:
                r.step()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        print("pipelines %d x %d frames/step: %.0f frames/s (%.3f ms per step of one pipeline)" % (n_active, B, n_active * B * steps / dt, dt / steps * 1e3))


if __name__ == "__main__":
    main()
