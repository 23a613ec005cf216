#!/usr/bin/env python3
"""bench.py -- frames/sec of the MI355X ORB front-end (BASELINE.json metric).

A "step" is one pass of the hot path (extract [+ SearchByProjection match]) over one batch of
synthetic 752x480 frames that are already resident in HBM.  One process per GPU; for N > 1 launch
with torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE from the env): every rank extracts its
own shard of frames (weak scaling, no data-path collective) and the per-step results are gathered
with one RCCL all_gather (the "trivial descriptor gather" of BASELINE.json config 4).

Prints ONE JSON line on rank 0 (see the driver contract in the task statement), including
  roofline     -- dominant kernel: algorithmic bytes per launch / HIP-event duration vs 8 TB/s
  cpu_baseline -- the CPU oracle (oracle/, a port of the reference algorithm) timed on host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (nFeatures, nFast, scale, levels, iniTh, minTh, W, H)   [SURVEY.md section 8 S0 defaults]
    "euroc_752x480": (1000, 40000, 1.2, 8, 20, 7, 752, 480),
    "batched_1280x720": (2000, 100000, 1.2, 8, 20, 7, 1280, 720),
    "tumvi_1024x1024": (1500, 100000, 1.2, 12, 20, 7, 1024, 1024),
}


def algorithmic_bytes_per_frame(ex, n_kp):
    """B = 5P - px0 - px_{L-1} + 1317 N (compulsory traffic, materialised pyramids)."""
    px = ex.levelW.astype(np.int64) * ex.levelH.astype(np.int64)
    P = int(px.sum())
    per_stage = {
        "pyramid_resize": int((P - px[-1]) + (P - px[0])),  # read every level but the last, write all but level 0
        # fused kernel: the level tile is read ONCE for both FAST and the Gaussian, blurred level written
        # (SURVEY's unfused figure is 3P: blur reads P + writes P, FAST reads P) -- we credit only 2P
        "fast_nms_blur": 2 * P,
        "quadtree": 0,
        "orient_brief": int((749 + 512 + 56) * n_kp),
    }
    return per_stage, 4 * P - int(px[0]) - int(px[-1]) + 1317 * int(n_kp)


def cpu_baseline(args_tuple, frames, budget_s):
    import oracle_py as O
    ref = O.Extractor(*args_tuple)
    ref.extract(frames[0])  # warm
    t0 = time.perf_counter()
    n = 0
    while n < len(frames) and (time.perf_counter() - t0 < budget_s or n < 3):
        ref.extract(frames[n])
        n += 1
    dt = time.perf_counter() - t0
    return n / dt, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per step per GPU")
    ap.add_argument("--workload", default="euroc_752x480", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import orbfe
    from orbfe import synth

    cfg = WORKLOADS[a.workload]
    W, H = cfg[6], cfg[7]
    B = a.batch
    ex = orbfe.ORBextractor(*cfg, device=local_rank, max_batch=B)
    cap = ex.cap

    # ---- synthetic stream, resident in HBM before the timed region ----
    frames = np.stack(list(synth.stream(W, H, B, index0=rank)))
    d_gray = torch.from_numpy(frames).to(dev)
    d_kp = torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    d_per = torch.zeros((B, ex.nlevels), dtype=torch.int32, device=dev)
    gather = world > 1 and not a.no_gather
    if gather:
        pack = torch.zeros((B, cap, 56), dtype=torch.uint8, device=dev)
        g_out = torch.zeros((world, B, cap, 56), dtype=torch.uint8, device=dev)
        g_n = torch.zeros((world, B), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step():
        ex.extract_batch_device(d_gray.data_ptr(), W * H, W, B, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(),
                                d_per.data_ptr(), stream.cuda_stream)
        if gather:
            pack[:, :, :24] = d_kp
            pack[:, :, 24:] = d_desc
            dist.all_gather_into_tensor(g_out, pack)
            dist.all_gather_into_tensor(g_n, d_n)

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize(dev)
    ex.set_stage_timing(True)  # HIP events on the launch stream, inside the timed region
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize(dev)
    barrier()
    dt = time.perf_counter() - t0
    stage_ms, ncalls = ex.stage_ms()
    ex.set_stage_timing(False)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    n_kp_mean = float(d_n.float().mean().item())
    if rank == 0:
        per_stage_bytes, b_frame = algorithmic_bytes_per_frame(ex, n_kp_mean)
        kernel_stages = [s for s in stage_ms if s != "total"]
        dom = max(kernel_stages, key=lambda s: stage_ms[s])
        dom_ms = stage_ms[dom] / max(1, ncalls)
        dom_bytes = per_stage_bytes[dom] * B
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        total_ms = stage_ms["total"] / max(1, ncalls)
        out = {
            "metric": "frames/sec ORB extract+match, 752x480 8-level 1000-feat; bit-exact kp/desc",
            "value": world * B * a.steps / dt,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "%s extract-only, %d frames/step/GPU resident in HBM, nFeatures=%d levels=%d "
                                   "scale=%.1f FAST %d/%d nFast=%d" % (a.workload, B, cfg[0], cfg[3], cfg[2], cfg[4],
                                                                         cfg[5], cfg[1]),
                       "frames_per_step": B * world, "mean_keypoints_per_frame": n_kp_mean,
                       "gather": "rccl all_gather of kp+desc per step" if gather else "none"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel_ms_per_launch": dom_ms, "algorithmic_bytes_per_launch": dom_bytes,
                         "pipeline_achieved_GBs": b_frame * B / (total_ms * 1e-3) / 1e9 if total_ms > 0 else 0.0,
                         "stage_ms_per_step": {k: v / max(1, ncalls) for k, v in stage_ms.items()}},
        }
        if not a.no_cpu_baseline:
            nsample = 64
            fps, n = cpu_baseline(cfg, frames[:nsample], a.cpu_seconds)
            out["cpu_baseline"] = {"value": fps, "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": "%d frames of the same stream through the single-thread C oracle "
                                             "(extract only), host %s, nproc=%d" % (n, os.uname().machine, os.cpu_count())}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
