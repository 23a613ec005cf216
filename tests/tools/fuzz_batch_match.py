#!/usr/bin/env python3
"""Randomised parity sweep of the BATCHED SearchByProjection path (orbfe_match_projection_batch_device: the throughput
kernels -- thread-per-map-point top-K, block-wide resolve with lazy re-evaluation): random batch sizes (>= 128 top-K blocks,
so that the large-launch kernels are selected), frame classes (default scene and the hostile look-alike classes), grids,
radii, ratios, map-point counts; every frame of the batch against the oracle, exact match indices and counts.
usage: fuzz_batch_match.py [n] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import match_scenarios as S  # noqa: E402
import oracle_py as O  # noqa: E402
import orbfe  # noqa: E402
from orbfe import synth  # noqa: E402

NAMES_O = ("projX", "projY", "viewCos", "trackDepth", "level", "inView", "bad", "observations")


def one(rng, k):
    W, H = int(rng.choice([320, 376, 640, 752])), int(rng.choice([240, 240, 400, 480]))
    levels = int(rng.integers(2, 9))
    nfeat = int(rng.choice([300, 1000, 1000, 1500]))
    cfg = (nfeat, 40000, 1.2, levels, 20, 7, W, H)
    M = int(rng.choice([600, 2000, 2000, 3000]))
    B = int(max(2, -(-128 // ((M + 255) // 256)) + rng.integers(0, 24)))  # >= 128 blocks of 256 map points in the launch
    kind = str(rng.choice(["default", "default", "noise", "plateau", "checker3", "lowtex"]))
    frames = [synth.frame(W, H, 3000 + 50 * k + b) if kind == "default" else synth.hostile(kind, W, H, 3000 + 50 * k + b) for b in range(B)]
    e = O.Extractor(*cfg)
    ex = orbfe.ORBextractor(*cfg, device=0, max_batch=B)
    cap = ex.cap
    cols, rows = int(rng.choice([16, 64, 64, 100])), int(rng.choice([12, 48, 48, 75]))
    th = float(rng.choice([3.0, 6.0, 20.0, 40.0]))
    nn = float(rng.choice([0.6, 0.75, 0.85, 1.0]))
    use_obs = bool(rng.random() < 0.5)
    refs, mps_all, mpd_all, obs_all = [], np.zeros((B, M), orbfe.MP_DTYPE), np.zeros((B, M, 32), np.uint8), np.full((B, cap), -1, np.int32)
    res = ex.extract_batch(frames)
    for b in range(B):
        kp, desc, _ = res[b]
        kpo = kp.view(O.KP_DTYPE)
        if len(kp) == 0:
            refs.append((0, np.zeros(0, np.int32)))
            continue
        mps, mpd, init_obs = S.projection_scenario(kpo, desc, M, int(rng.integers(1 << 30)), O.MP_DTYPE, NAMES_O, e.nLevels)
        if not use_obs:
            init_obs = None
        else:
            obs_all[b, :len(kp)] = init_obs
        mps_all[b], mpd_all[b] = mps.view(orbfe.MP_DTYPE), mpd
        fvo = O.make_frame_view(kpo, desc, cols, rows, 0.0, 0.0, float(W), float(H), e.scaleFactors)
        refs.append(O.search_by_projection(fvo, mps, mpd, init_obs, th, nn))
    dev = torch.device("cuda", 0)
    d_kp = torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    for b in range(B):
        kp, desc, _ = res[b]
        d_kp[b, :len(kp)] = torch.from_numpy(kp.view(np.uint8).reshape(-1, 24)).to(dev)
        d_desc[b, :len(kp)] = torch.from_numpy(desc).to(dev)
        d_n[b] = len(kp)
    d_mps = torch.from_numpy(mps_all.view(np.uint8).reshape(-1)).to(dev)
    d_mpd = torch.from_numpy(mpd_all.reshape(-1)).to(dev)
    d_obs = torch.from_numpy(obs_all.reshape(-1)).to(dev) if use_obs else None
    d_match = torch.full((B, cap), -7, dtype=torch.int32, device=dev)
    d_nm = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    orbfe.ORBmatcher(ex).SearchByProjection_batch_device(B, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, cols, rows, 0.0, 0.0,
                                                         float(W), float(H), M, d_mps.data_ptr(), d_mpd.data_ptr(),
                                                         d_obs.data_ptr() if use_obs else None, th, nn, d_match.data_ptr(), d_nm.data_ptr())
    torch.cuda.synchronize()
    match, nm = d_match.cpu().numpy(), d_nm.cpu().numpy()
    for b in range(B):
        n_ref, out_ref = refs[b]
        k = len(res[b][0])
        if k == 0:
            continue
        assert nm[b] == n_ref and np.array_equal(match[b, :k], out_ref), (
            "batched SearchByProjection differs: case %d frame %d kind %s cfg %s grid %dx%d th %g nn %g M %d B %d obs %s: %d vs %d matches" % (
                k, b, kind, cfg, cols, rows, th, nn, M, B, use_obs, nm[b], n_ref))
    return B, kind


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    frames = 0
    kinds = {}
    for k in range(n):
        B, kind = one(rng, k)
        frames += B
        kinds[kind] = kinds.get(kind, 0) + 1
        print("case %d ok (%d frames, %s)" % (k, B, kind), flush=True)
    print("fuzz_batch_match: %d cases, %d frames, all exact; image classes %s" % (n, frames, kinds))
