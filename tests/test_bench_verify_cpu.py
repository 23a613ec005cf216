"""bench.py's `verified` field is only worth something if the checker itself catches a wrong batch: drive StepRunner on CPU tensors
with the ORACLE standing in for the HIP calls (same buffer layout), then ask verify_last_step -- it must accept the untouched
buffers and reject a flipped descriptor bit, a moved keypoint, a wrong count and a wrong match index."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "orb_slam3_v1.0_amd", "python"))


def test_verify_last_step_accepts_the_oracle_and_rejects_corruption():
    import bench
    import oracle_py as O
    import orbfe
    from orbfe import synth
    cfg = (150, 6000, 1.2, 3, 20, 7, 160, 120)
    W, H = cfg[6], cfg[7]
    B, M, n_sets = 5, 60, 2
    e = O.Extractor(*cfg)
    cap = e.cap
    frames = [np.stack(list(synth.stream(W, H, B, index0=100 * s))) for s in range(n_sets)]
    rng = np.random.default_rng(3)
    state = {"mps": [], "mpd": []}
    ext = {}
    for fs in range(n_sets):  # map points per frame set (bench.make_points), from the oracle's own extraction
        mps_all, mpd_all = np.zeros((B, M), orbfe.MP_DTYPE), np.zeros((B, M, 32), np.uint8)
        for i in range(B):
            kp, desc, _ = e.extract(frames[fs][i])
            ext[(fs, i)] = (kp, desc)
            mps_all[i], mpd_all[i] = bench.make_map_points(kp, len(kp), desc, M, rng, e.nLevels, orbfe.MP_DTYPE)
        state["mps"].append(torch.from_numpy(mps_all.view(np.uint8).reshape(-1)))
        state["mpd"].append(torch.from_numpy(mpd_all.reshape(-1)))

    def extract_fn(b, fs):
        for i in range(B):
            kp, desc = ext[(fs, i)]
            b["n"][i] = len(kp)
            b["kp"][i, :len(kp)] = torch.from_numpy(kp.view(np.uint8).reshape(len(kp), 24))
            b["desc"][i, :len(kp)] = torch.from_numpy(desc)

    def match_fn(b, fs):
        mps = state["mps"][fs].numpy().view(orbfe.MP_DTYPE).reshape(B, M)
        mpd = state["mpd"][fs].numpy().reshape(B, M, 32)
        for i in range(B):
            kp, desc = ext[(fs, i)]
            fv = O.make_frame_view(kp, desc, bench.GRID[0], bench.GRID[1], 0.0, 0.0, float(W), float(H), e.scaleFactors)
            n, m = O.search_by_projection(fv, mps[i].view(O.MP_DTYPE), mpd[i], None, bench.MATCH_TH, bench.MATCH_NN)
            b["nmatch"][i] = n
            b["match"][i, :len(kp)] = torch.from_numpy(m)

    r = bench.StepRunner(torch.device("cpu"), B, cap, e.nLevels, extract_fn, match_fn, None, 1, False, True, frame_sets=n_sets)
    for _ in range(3):
        r.step()
    assert r.last[1] == 0  # the third step used frame set 0 again
    v = bench.verify_last_step(r, frames, state, B, cap, M, cfg)
    assert v["kp_desc_equal"] is True and v["match_equal"] is True and v["frames"] == B and v["frame_set"] == 0
    b = r.last[0]
    # every kind of damage is seen, and reported with the frame it sits in
    b["desc"][2, 3, 7] ^= 0x10
    v = bench.verify_last_step(r, frames, state, B, cap, M, cfg)
    assert v["kp_desc_equal"] is False and v["first_mismatch_frame"] == 2
    b["desc"][2, 3, 7] ^= 0x10
    b["kp"][4, 0, 0] ^= 1  # lowest mantissa bit of a keypoint's x
    assert bench.verify_last_step(r, frames, state, B, cap, M, cfg)["kp_desc_equal"] is False
    b["kp"][4, 0, 0] ^= 1
    b["n"][1] -= 1
    assert bench.verify_last_step(r, frames, state, B, cap, M, cfg)["kp_desc_equal"] is False
    b["n"][1] += 1
    hit = int(torch.nonzero(b["match"][0, :int(b["n"][0])] >= 0)[0])
    b["match"][0, hit] += 1
    v = bench.verify_last_step(r, frames, state, B, cap, M, cfg)
    assert v["kp_desc_equal"] is True and v["match_equal"] is False and v["first_mismatch_frame"] == 0
    b["match"][0, hit] -= 1
    v = bench.verify_last_step(r, frames, state, B, cap, M, cfg)
    assert v["kp_desc_equal"] and v["match_equal"] and "first_mismatch_frame" not in v
    # extract-only lines carry no match verdict
    r2 = bench.StepRunner(torch.device("cpu"), B, cap, e.nLevels, extract_fn, None, None, 1, False, True, frame_sets=n_sets)
    r2.step()
    v2 = bench.verify_last_step(r2, frames, state, B, cap, 0, cfg)
    assert v2["kp_desc_equal"] is True and "match_equal" not in v2
