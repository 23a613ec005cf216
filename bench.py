#!/usr/bin/env python3
"""bench.py -- frames/sec of the MI355X ORB front-end (BASELINE.json metric).

A "step" is one pass of the hot path -- ORB extraction of a batch of synthetic 752x480 frames that
are already resident in HBM, followed by SearchByProjection of 2000 map points per frame against
the fresh keypoints (SURVEY.md section 8d, config C3 recipe).  One process per GPU; for N > 1 launch with
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE from the env): every rank works on its own
shard of frames (weak scaling, no data-path collective) and the per-step results are gathered with
one RCCL all_gather (the "trivial descriptor gather" of BASELINE.json config 4).

Prints ONE JSON line on rank 0 (driver contract), including
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

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
HBM_MEASURED_COPY_GBS = 6290.0  # same guide: measured copy peak

WORKLOADS = {
    # name: (nFeatures, nFast, scale, levels, iniTh, minTh, W, H)   [SURVEY.md section 8 S0 defaults]
    "euroc_752x480": (1000, 40000, 1.2, 8, 20, 7, 752, 480),
    "batched_1280x720": (2000, 100000, 1.2, 8, 20, 7, 1280, 720),
    "tumvi_1024x1024": (1500, 100000, 1.2, 12, 20, 7, 1024, 1024),
}
GRID = (64, 48)          # mFrameGridCols x mFrameGridRows (mono_inertial_node.cpp:187-188)
MATCH_TH, MATCH_NN = 20.0, 0.85  # Tracking.cc:1108-1113 before IMU init
N_MAP_POINTS = 2000


def algorithmic_bytes_per_frame(ex, n_kp):
    """Compulsory HBM traffic per frame with materialised pyramids (minus the P
    bytes saved by reading each level once for both FAST and the Gaussian)."""
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


STAGE_KERNEL = {"pyramid_resize": "resize_kernel", "fast_nms_blur": "fast_blur_kernel", "quadtree": "quadtree_kernel",
                "orient_brief": "orient_brief_kernel"}


def pmc_traffic(stage, workload, batch):
    """HBM bytes per launch of the stage's kernel from the committed PMC passes (profiles/r01_traffic_pmc.json:
    FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc runs, gfx950 corrections applied by
    tools/traffic_report.py).  Counters cannot be read from inside the timed process, so this is the number of the
    last profiled run of the SAME workload / batch; null otherwise."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic_pmc.json")))
        meta = d.get("_meta", {})
        if meta.get("workload") != workload or meta.get("frames_per_launch") != batch:
            return None
        tot = [v["hbm_bytes_per_launch"] for k, v in d["kernels"].items() if STAGE_KERNEL[stage] in k]
        return float(tot[0]) if tot else None  # average over the profiled launches of that kernel
    except (OSError, ValueError, KeyError):
        return None


def make_map_points(kp, n, desc, M, rng, n_levels, mp_dtype):
    """C3 recipe, vectorised: descriptor of a random keypoint with 0..20 bit flips, projection = that
    keypoint +- 3 px, level = its octave."""
    src = rng.integers(0, max(n, 1), M)
    mps = np.zeros(M, mp_dtype)
    mps["proj_x"] = kp["x"][src] + rng.uniform(-3, 3, M).astype(np.float32)
    mps["proj_y"] = kp["y"][src] + rng.uniform(-3, 3, M).astype(np.float32)
    mps["view_cos"] = 1.0
    mps["track_depth"] = 1.0
    mps["level"] = np.minimum(kp["octave"][src], n_levels - 1)
    mps["in_view"] = 1
    mps["observations"] = rng.integers(0, 4, M)
    d = desc[src].copy()
    nflip = rng.integers(0, 21, M)
    for j in range(20):
        act = np.nonzero(j < nflip)[0]
        pos = rng.integers(0, 256, len(act))
        np.bitwise_xor.at(d, (act, pos >> 3), (1 << (pos & 7)).astype(np.uint8))
    return mps, d


def host_cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args_tuple, frames, budget_s, with_match, mp_dtype):
    """Single-thread CPU oracle on a bounded sample of the same stream (extract [+ match])."""
    import oracle_py as O
    ref = O.Extractor(*args_tuple)
    W, H = args_tuple[6], args_tuple[7]
    rng = np.random.default_rng(7)
    ref.extract(frames[0])  # warm
    t0 = time.perf_counter()
    n = 0
    t_gen = 0.0
    while n < len(frames) and (time.perf_counter() - t0 - t_gen < budget_s or n < 3):
        kp, desc, _ = ref.extract(frames[n])
        if with_match and len(kp):
            tg = time.perf_counter()
            mps, mpd = make_map_points(kp, len(kp), desc, N_MAP_POINTS, rng, ref.nLevels, mp_dtype)
            fv = O.make_frame_view(kp, desc, GRID[0], GRID[1], 0.0, 0.0, float(W), float(H), ref.scaleFactors)
            t_gen += time.perf_counter() - tg  # input synthesis is not part of the measured path
            O.search_by_projection(fv, mps.view(O.MP_DTYPE), mpd, None, MATCH_TH, MATCH_NN)
        n += 1
    dt = time.perf_counter() - t0 - t_gen
    return n / dt, n


def cpu_baseline_all_cores(args_tuple, frames, budget_s, with_match, mp_dtype, threads):
    """The same oracle path with the sample's frames dealt to `threads` host threads (ctypes releases the GIL inside
    the C calls; the map-point synthesis in between is Python and is excluded per thread like above)."""
    from concurrent.futures import ThreadPoolExecutor
    per = max(1, len(frames) // threads)
    chunks = [frames[i * per:(i + 1) * per] for i in range(threads) if len(frames[i * per:(i + 1) * per])]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(len(chunks)) as ex:
        res = list(ex.map(lambda c: cpu_baseline(args_tuple, c, budget_s, with_match, mp_dtype), chunks))
    wall = time.perf_counter() - t0
    n = sum(r[1] for r in res)
    # aggregate rate = sum of the per-thread rates (each excludes its own input synthesis); wall is reported too
    return sum(r[0] for r in res), n, len(chunks), wall


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="frames per step per GPU (128: 131 k, 256: 142 k, 512: 149 k, 1024: 147 k frames/s)")
    ap.add_argument("--workload", default="euroc_752x480", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads of the all-cores CPU baseline (0/1 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the RCCL result gather even at N=1 (a 1-rank process group): exercises the N>1 code path on one GPU")
    ap.add_argument("--no-match", action="store_true", help="extract only")
    ap.add_argument("--no-overlap", action="store_true", help="match on the extraction stream (no 2-stream pipelining)")
    ap.add_argument("--lib", default=None, help="load this build of liborbfe instead (tools/: the timing-only ablation build)")
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
    # everything (our kernels, torch copies, the RCCL gather) is ordered on ONE explicit stream; the
    # default stream's handle is 0, which the C ABI reads as "use the handle's own stream"
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    dist = None
    if world > 1 or a.force_gather:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import orbfe
    from orbfe import synth
    if a.lib:
        orbfe.LIB_PATH = os.path.abspath(a.lib)

    cfg = WORKLOADS[a.workload]
    W, H = cfg[6], cfg[7]
    B = a.batch
    M = 0 if a.no_match else N_MAP_POINTS
    ex = orbfe.ORBextractor(*cfg, device=local_rank, max_batch=B)
    matcher = orbfe.ORBmatcher(ex)
    cap = ex.cap

    # ---- synthetic stream, resident in HBM before the timed region ----
    frames = np.stack(list(synth.stream(W, H, B, index0=rank)))
    d_gray = torch.from_numpy(frames).to(dev)
    # double-buffered outputs: extraction of step i+1 (stream s1) overlaps the matching of step i (stream s2)
    nbuf = 2 if (M and not a.no_overlap) else 1
    bufs = []
    for _ in range(nbuf):
        bufs.append(dict(kp=torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev),
                         desc=torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
                         n=torch.zeros(B, dtype=torch.int32, device=dev),
                         per=torch.zeros((B, ex.nlevels), dtype=torch.int32, device=dev),
                         match=torch.full((B, cap), -1, dtype=torch.int32, device=dev),
                         nmatch=torch.zeros(B, dtype=torch.int32, device=dev),
                         ev_ext=torch.cuda.Event(), ev_done=torch.cuda.Event()))
    gather = (world > 1 or a.force_gather) and not a.no_gather
    if gather:
        pack = torch.zeros((B, cap, 60), dtype=torch.uint8, device=dev)  # kp 24 + desc 32 + match 4
        g_out = torch.zeros((world * B, cap, 60), dtype=torch.uint8, device=dev)  # rank-major concatenation
        g_n = torch.zeros(world * B, dtype=torch.int32, device=dev)
    s1 = torch.cuda.current_stream(dev)
    s2 = torch.cuda.Stream(dev) if nbuf == 2 else s1

    def extract(b):
        ex.extract_batch_device(d_gray.data_ptr(), W * H, W, B, b["kp"].data_ptr(), b["desc"].data_ptr(),
                                b["n"].data_ptr(), b["per"].data_ptr(), s1.cuda_stream)

    # ---- map points (C3 recipe) from one untimed extraction; resident in HBM as well ----
    extract(bufs[0])
    torch.cuda.synchronize(dev)
    if M:
        kp_h = bufs[0]["kp"].cpu().numpy().reshape(B, cap * 24).view(orbfe.KP_DTYPE).reshape(B, cap)
        desc_h = bufs[0]["desc"].cpu().numpy()
        n_h = bufs[0]["n"].cpu().numpy()
        rng = np.random.default_rng(1234 + rank)
        mps_all = np.zeros((B, M), orbfe.MP_DTYPE)
        mpd_all = np.zeros((B, M, 32), np.uint8)
        for b in range(B):
            mps_all[b], mpd_all[b] = make_map_points(kp_h[b], int(n_h[b]), desc_h[b], M, rng, ex.nlevels, orbfe.MP_DTYPE)
        d_mps = torch.from_numpy(mps_all.view(np.uint8).reshape(-1)).to(dev)
        d_mpd = torch.from_numpy(mpd_all.reshape(-1)).to(dev)
    ev_m0 = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)]
    ev_m1 = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)]
    counter = [0]

    def step(i=None):
        b = bufs[counter[0] % nbuf]
        counter[0] += 1
        if nbuf == 2:
            s1.wait_event(b["ev_done"])  # this buffer's previous match (two steps ago) must have finished
        extract(b)
        if nbuf == 2:
            b["ev_ext"].record(s1)
            s2.wait_event(b["ev_ext"])
        if M:
            if i is not None:
                ev_m0[i].record(s2)
            matcher.SearchByProjection_batch_device(B, b["kp"].data_ptr(), b["desc"].data_ptr(), b["n"].data_ptr(), cap,
                                                    GRID[0], GRID[1], 0.0, 0.0, float(W), float(H), M, d_mps.data_ptr(),
                                                    d_mpd.data_ptr(), None, MATCH_TH, MATCH_NN, b["match"].data_ptr(),
                                                    b["nmatch"].data_ptr(), stream=s2.cuda_stream)
            if i is not None:
                ev_m1[i].record(s2)
        if gather:
            with torch.cuda.stream(s2):
                pack[:, :, :24] = b["kp"]
                pack[:, :, 24:56] = b["desc"]
                pack[:, :, 56:] = b["match"].view(torch.uint8).reshape(B, cap, 4)
                dist.all_gather_into_tensor(g_out, pack)
                dist.all_gather_into_tensor(g_n, b["n"])
        if nbuf == 2:
            b["ev_done"].record(s2)

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
    for i in range(a.steps):
        step(i)
    torch.cuda.synchronize(dev)
    barrier()
    dt = time.perf_counter() - t0
    stage_ms, ncalls = ex.stage_ms()
    ex.set_stage_timing(False)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    n_kp_mean = float(bufs[0]["n"].float().mean().item())
    if rank == 0:
        per_stage_bytes, b_frame = algorithmic_bytes_per_frame(ex, n_kp_mean)
        stage_avg = {k: v / max(1, ncalls) for k, v in stage_ms.items()}
        if M:
            stage_avg["match_projection"] = sum(e0.elapsed_time(e1) for e0, e1 in zip(ev_m0, ev_m1)) / a.steps
        kernel_stages = [s for s in per_stage_bytes]
        dom = max(kernel_stages, key=lambda s: stage_avg[s])
        dom_ms = stage_avg[dom]
        dom_bytes = per_stage_bytes[dom] * B
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        total_ms = stage_avg["total"]
        what = "extract-only" if not M else "extract + SearchByProjection(%d map points/frame, grid %dx%d, th=%g, nnRatio=%g)" % (
            M, GRID[0], GRID[1], MATCH_TH, MATCH_NN)
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
            "config": {"workload": "%s synthetic stream, %s, %d frames/step/GPU resident in HBM, nFeatures=%d levels=%d "
                                   "scale=%.1f FAST %d/%d nFast=%d" % (a.workload, what, B, cfg[0], cfg[3], cfg[2], cfg[4],
                                                                         cfg[5], cfg[1]),
                       "frames_per_step": B * world, "mean_keypoints_per_frame": n_kp_mean,
                       "mean_matches_per_frame": float(bufs[0]["nmatch"].float().mean().item()) if M else None,
                       "gather": "rccl all_gather of kp+desc+match per step" if gather else "none",
                       "streams": "extract(step i+1) || match(step i), double-buffered outputs" if nbuf == 2 else "single stream"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_peak": achieved / HBM_MEASURED_COPY_GBS,
                         "traffic": pmc_traffic(dom, a.workload, B),
                         "kernel_ms_per_launch": dom_ms, "algorithmic_bytes_per_launch": dom_bytes,
                         "extract_pipeline_achieved_GBs": b_frame * B / (total_ms * 1e-3) / 1e9 if total_ms > 0 else 0.0,
                         "stage_ms_per_step": stage_avg},
        }
        if not a.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            fps, n = cpu_baseline(cfg, frames[:64], a.cpu_seconds, bool(M), orbfe.MP_DTYPE)
            out["cpu_baseline"] = {"value": fps, "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": "%d frames of the same stream through the single-thread C oracle (%s), "
                                             "host CPU %s, nproc=%d" % (n, what.split("(")[0].strip(), host_cpu_model(), os.cpu_count())}
            if a.cpu_threads > 1:
                fps_all, n_all, used, wall = cpu_baseline_all_cores(cfg, frames[:min(len(frames), 8 * a.cpu_threads)],
                                                                    a.cpu_seconds, bool(M), orbfe.MP_DTYPE, a.cpu_threads)
                out["cpu_baseline"]["all_cores"] = {"value": fps_all, "unit": "frames/s", "cores": used,
                                                    "sample": "%d frames over %d host threads, %.1f s wall" % (n_all, used, wall)}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
