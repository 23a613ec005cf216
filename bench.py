#!/usr/bin/env python3
"""bench.py -- frames/sec of the MI355X ORB front-end (BASELINE.json metric).

A "step" is one pass of the hot path -- ORB extraction of a batch of frames that are already resident in HBM, followed
by SearchByProjection of 2000 map points per frame against the fresh keypoints (SURVEY.md section 8d, config C3 recipe).
One process per GPU.  `python bench.py --gpus N` is self-contained: with N > 1 and no WORLD_SIZE in the environment the
parent -- before it touches the GPU -- starts N ranks through torch.distributed.run on 127.0.0.1 and relays rank 0's
JSON line; under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the env) it is one of those ranks.
`n_gpus` is the size of the RCCL group that actually ran, and a mismatch with --gpus is an error.
Two multi-GPU modes (`config.workload` names the one that ran):
  weak    every rank brings its own --batch frames (default for the 752x480 headline workload); per step the ranks
          exchange only the per-frame keypoint / match COUNTS (8 B per frame) -- results stay on the GPU that made them;
  strong  BASELINE config 4: --workload batched_1280x720 is 512 frames IN TOTAL, rank r owns shard_range(512, r, N)
          (64 per GPU at N = 8) and the padded results (keypoints + descriptors + match indices, 60 B per slot) are
          all-gathered once per step -- the "trivial descriptor gather" of BASELINE.json config 4.  `value` is still
          total frames / time.  --emulate-world K runs rank 0's shard of a K-rank split on ONE GPU (no collective).
There is no data-path collective; `config.gather_bytes_per_step` says what each GPU receives per step.

Prints ONE JSON line on rank 0 (driver contract), including
  roofline      dominant kernel: algorithmic bytes per launch / HIP-event duration vs 8 TB/s, plus the instruction
                side (`issue`: priced at the clock MEASURED in this run) and the PMC traffic / VALU-busy share of the committed
                profile of the SAME sources (`profiles_head`)
  sustained     a second timed region behind the driver's K steps: the same step loop for >= 5 s, every step's duration from HIP
                events (p50 / p99 / max), the first and the last 100 steps, and the shader clock the chip held meanwhile, read in
                the kernel (s_memtime over s_memrealtime, orbfe_debug_clock_probe) on a stream beside the workload
  verified      8 frames of the LAST timed step's device buffers re-computed by the CPU oracle: keypoints + descriptors and the
                match indices must be equal byte for byte, or the exit code is non-zero
  cpu_baseline  the CPU oracle (oracle/, a port of the reference algorithm) pinned to one host core
  latency       the reference's own call shape -- ONE host frame per call (src/Frame.cc:178-189): orbfe_extract from
                pageable and pinned memory, orbfe_match_projection (2000 map points), orbfe_prepare_and_extract, each
                the median of >= 200 calls, next to the single-thread oracle on the same call
  value_host_io frames/s with host pointers in and out through the pipelined ring (orbfe_stream_*)
  matcher       map points/s and brute-force-equivalent Hamming pairs/s of the SearchByProjection stage.
--images DIR runs the same step on grey-converted image files (PGM / PNG / JPEG via PIL, centre-cropped or resized to
the workload geometry) for BASELINE configs 2/3/5 where EuRoC / TUM-VI data exists; the default is synthetic.
"""
import argparse
import contextlib
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
HBM_MEASURED_COPY_GBS = 6290.0  # same guide: measured copy peak
SIMDS, CLK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMD-32; 2.4 GHz = max clock (the fallback when no clock could be measured)

WORKLOADS = {
    # name: (nFeatures, nFast, scale, levels, iniTh, minTh, W, H)   [SURVEY.md section 8 S0 defaults]
    "euroc_752x480": (1000, 40000, 1.2, 8, 20, 7, 752, 480),
    # SURVEY.md section 8, SPEC DECISION S0: nFastFeatures = 16 x nFeatures (the default above is 40 x: no cap ever fires)
    "euroc_752x480_s0": (1000, 16000, 1.2, 8, 20, 7, 752, 480),
    # the reference node's own ratio nFastFeatures = 1.6 x nFeatures (ros2_ws/src/mono-inertial/src/mono_inertial_node.cpp:88-89:
    # 10000 / 16000): the per-level FAST cap fires on every level that holds more corners than that
    "euroc_752x480_node": (1000, 1600, 1.2, 8, 20, 7, 752, 480),
    "batched_1280x720": (2000, 100000, 1.2, 8, 20, 7, 1280, 720),
    "tumvi_1024x1024": (1500, 100000, 1.2, 12, 20, 7, 1024, 1024),
}
C4_TOTAL_FRAMES = 512    # BASELINE.json configs[3]: 512 frames 1280x720 in total, sharded over the GPUs
GRID = (64, 48)          # mFrameGridCols x mFrameGridRows (mono_inertial_node.cpp:187-188)
MATCH_TH, MATCH_NN = 20.0, 0.85  # Tracking.cc:1108-1113 before IMU init
N_MAP_POINTS = 2000
SLOT_BYTES = 60          # gathered record: keypoint 24 + descriptor 32 + match index 4


def algorithmic_bytes_per_frame(level_w, level_h, n_kp):
    """Compulsory HBM traffic per frame with materialised pyramids (minus the P
    bytes saved by reading each level once for both FAST and the Gaussian)."""
    px = np.asarray(level_w, np.int64) * np.asarray(level_h, np.int64)
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


STAGE_KERNEL = {"pyramid_resize": "pyramid_kernel", "fast_nms_blur": "fast_blur_kernel", "quadtree": "quadtree_kernel",
                "orient_brief": "orient_brief_kernel"}


def profile_for(stage, workload, frames_per_launch):
    """Counter-derived numbers of the stage's kernel from the newest committed profile summary
    (profiles/*_kernels.json, tools/profile_all.sh) -- but only if that profile was taken on THESE sources (content
    hash of csrc/) with the same workload and launch size; otherwise None: counters cannot be read from inside the
    timed process, and a number of other code would be a wrong number."""
    from orbfe.provenance import source_sha
    try:
        sha = source_sha()
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_kernels.json")), reverse=True):
            d = json.load(open(f))
            meta = d.get("_meta", {})
            run = meta.get("bench_plain") or meta.get("trace") or {}
            if meta.get("source_sha") != sha or run.get("workload") != workload or run.get("frames_per_step") != frames_per_launch:
                continue
            for k, v in d.get("kernels", {}).items():
                if STAGE_KERNEL[stage] in k:
                    return {"file": os.path.basename(f), "git_head": meta.get("git_head"), **v}
    except (OSError, ValueError, KeyError):
        pass
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


def cpu_baseline(args_tuple, frames, budget_s, with_match, mp_dtype, pin_core=None, warmup=5, min_frames=30):
    """Single-thread CPU oracle on a bounded sample of the same stream (extract [+ match]), SURVEY.md section 8d:
    pinned to one core, `warmup` untimed frames, then per-frame times of >= `min_frames` frames (budget permitting);
    the rate is 1 / MEDIAN frame time.  Returns (frames/s, timed frames)."""
    import oracle_py as O
    old_aff = None
    if pin_core is not None and hasattr(os, "sched_setaffinity"):
        try:
            old_aff = os.sched_getaffinity(0)
            os.sched_setaffinity(0, {pin_core if pin_core in old_aff else min(old_aff)})
        except OSError:
            old_aff = None
    try:
        ref = O.Extractor(*args_tuple)
        W, H = args_tuple[6], args_tuple[7]
        rng = np.random.default_rng(7)

        def one(img):
            t0 = time.perf_counter()
            kp, desc, _ = ref.extract(img)
            dt = time.perf_counter() - t0
            if with_match and len(kp):
                # input synthesis is not part of the measured path
                mps, mpd = make_map_points(kp, len(kp), desc, N_MAP_POINTS, rng, ref.nLevels, mp_dtype)
                fv = O.make_frame_view(kp, desc, GRID[0], GRID[1], 0.0, 0.0, float(W), float(H), ref.scaleFactors)
                t1 = time.perf_counter()
                O.search_by_projection(fv, mps.view(O.MP_DTYPE), mpd, None, MATCH_TH, MATCH_NN)
                dt += time.perf_counter() - t1
            return dt

        for i in range(min(warmup, len(frames))):
            one(frames[i])
        times, t_start = [], time.perf_counter()
        i = 0
        while i < max(min_frames, len(frames)) and (len(times) < 3 or time.perf_counter() - t_start < budget_s):
            times.append(one(frames[(warmup + i) % len(frames)]))
            i += 1
        return 1.0 / float(np.median(times)), len(times)
    finally:
        if old_aff is not None:
            os.sched_setaffinity(0, old_aff)


def cpu_baseline_all_cores(args_tuple, frames, budget_s, with_match, mp_dtype, threads):
    """The same oracle path with the sample's frames dealt to `threads` host threads (ctypes releases the GIL inside
    the C calls; the map-point synthesis in between is Python and is excluded per thread like above)."""
    from concurrent.futures import ThreadPoolExecutor
    per = max(1, len(frames) // threads)
    chunks = [frames[i * per:(i + 1) * per] for i in range(threads) if len(frames[i * per:(i + 1) * per])]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(len(chunks)) as ex:
        res = list(ex.map(lambda c: cpu_baseline(args_tuple, c, budget_s, with_match, mp_dtype, None, 1, 3), chunks))
    wall = time.perf_counter() - t0
    n = sum(r[1] for r in res)
    # aggregate rate = sum of the per-thread rates (each excludes its own input synthesis); wall is reported too
    return sum(r[0] for r in res), n, len(chunks), wall


class StepRunner:
    """One bench step on one rank: extract(step i+1) on stream s1 overlapped with match(step i) on stream s2 through
    double-buffered outputs, then the per-step RCCL gather of the padded results.  The device work goes through two
    callables so that the N > 1 path (buffer rotation, packing, all_gather_into_tensor, rank-major layout) also runs
    under gloo on CPU with a stub extractor (tests/test_distributed_cpu.py):
        extract_fn(buf, frame_set)  fills buf["kp"|"desc"|"n"|"per"]      (enqueued on s1)
        match_fn(buf, frame_set)    fills buf["match"|"nmatch"]            (enqueued on s2), or None
    """

    def __init__(self, device, B, cap, n_levels, extract_fn, match_fn, dist=None, world=1, gather=False, overlap=True,
                 frame_sets=1):
        self.dev, self.B, self.cap, self.world = device, B, cap, world
        self.extract_fn, self.match_fn, self.dist = extract_fn, match_fn, dist
        self.frame_sets = frame_sets
        self.cuda = device.type == "cuda"
        self.nbuf = 2 if (match_fn is not None and overlap and self.cuda) else 1
        self.bufs = []
        for _ in range(self.nbuf):
            self.bufs.append(dict(kp=torch.zeros((B, cap, 24), dtype=torch.uint8, device=device),
                                  desc=torch.zeros((B, cap, 32), dtype=torch.uint8, device=device),
                                  n=torch.zeros(B, dtype=torch.int32, device=device),
                                  per=torch.zeros((B, n_levels), dtype=torch.int32, device=device),
                                  match=torch.full((B, cap), -1, dtype=torch.int32, device=device),
                                  nmatch=torch.zeros(B, dtype=torch.int32, device=device),
                                  ev_ext=torch.cuda.Event() if self.cuda else None,
                                  ev_done=torch.cuda.Event() if self.cuda else None))
        # gather: False / "none", "counts" (per-frame keypoint + match counts only) or True / "full" (padded results)
        self.gather = {True: "full", False: "none", None: "none"}.get(gather, gather)
        assert self.gather in ("none", "counts", "full")
        if self.gather == "full":
            self.pack = torch.zeros((B, cap, SLOT_BYTES), dtype=torch.uint8, device=device)
            self.g_out = torch.zeros((world * B, cap, SLOT_BYTES), dtype=torch.uint8, device=device)  # rank-major
            self.g_n = torch.zeros(world * B, dtype=torch.int32, device=device)
        elif self.gather == "counts":
            self.cnt = torch.zeros((B, 2), dtype=torch.int32, device=device)
            self.g_cnt = torch.zeros((world * B, 2), dtype=torch.int32, device=device)  # rank-major: (keypoints, matches)
        self.s1 = self.s2 = None
        if self.cuda:
            self.s1 = torch.cuda.current_stream(device)
            # BENCH_S2_PRIORITY: experiment knob (-1 = high priority for the matcher's stream); measured, no gain
            self.s2 = torch.cuda.Stream(device, priority=int(os.environ.get("BENCH_S2_PRIORITY", "0"))) if self.nbuf == 2 else self.s1
        self.count = 0
        self.match_events = []

    def _on(self, stream):
        return torch.cuda.stream(stream) if self.cuda else contextlib.nullcontext()

    def step(self, timed=False, done_event=None):
        """one step; done_event (optional, timing-enabled) is recorded where the step's last piece of work is enqueued"""
        b = self.bufs[self.count % self.nbuf]
        fs = self.count % self.frame_sets
        self.count += 1
        if self.nbuf == 2:
            self.s1.wait_event(b["ev_done"])  # this buffer's previous match (two steps ago) must have finished
        self.extract_fn(b, fs)
        if self.nbuf == 2:
            b["ev_ext"].record(self.s1)
            self.s2.wait_event(b["ev_ext"])
        if self.match_fn is not None:
            if timed and self.cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.s2)
            self.match_fn(b, fs)
            if timed and self.cuda:
                e1.record(self.s2)
                self.match_events.append((e0, e1))
        if self.gather == "full":
            with self._on(self.s2):
                self.pack[:, :, :24] = b["kp"]
                self.pack[:, :, 24:56] = b["desc"]
                self.pack[:, :, 56:] = b["match"].view(torch.uint8).reshape(self.B, self.cap, 4)
                self.dist.all_gather_into_tensor(self.g_out, self.pack)
                self.dist.all_gather_into_tensor(self.g_n, b["n"])
        elif self.gather == "counts":
            with self._on(self.s2):
                self.cnt[:, 0] = b["n"]
                self.cnt[:, 1] = b["nmatch"]
                self.dist.all_gather_into_tensor(self.g_cnt, self.cnt)
        if self.nbuf == 2:
            b["ev_done"].record(self.s2)
        if done_event is not None:
            done_event.record(self.s2)  # (s2 is s1 when the step runs on one stream)
        self.last = (b, fs)
        return b

    def verify_gather(self):
        """after the last step, stream idle: this rank's slots of the gathered arrays must hold exactly what it packed, and
        every rank must hold the same gathered bytes (a byte sum compared through MAX / MIN all-reduces).  -> (slots of this
        rank that differ, gathered bytes identical on all ranks) or None when no collective ran"""
        if self.gather == "none":
            return None
        r = self.dist.get_rank() if self.dist is not None else 0
        lo, hi = r * self.B, (r + 1) * self.B
        b = self.last[0]
        if self.gather == "full":
            want = torch.cat([b["kp"], b["desc"], b["match"].view(torch.uint8).reshape(self.B, self.cap, 4)], dim=2)
            bad = int((self.g_out[lo:hi] != want).any(dim=2).sum().item()) + int((self.g_n[lo:hi] != b["n"]).sum().item())
            digest = self.g_out.sum(dtype=torch.int64) + 131 * self.g_n.sum(dtype=torch.int64)
        else:
            want = torch.stack([b["n"], b["nmatch"]], dim=1)
            bad = int((self.g_cnt[lo:hi] != want).any(dim=1).sum().item())
            digest = (self.g_cnt.to(torch.int64) * torch.tensor([1, 65537], dtype=torch.int64, device=self.dev)).sum()
        t = torch.stack([digest, -digest, torch.tensor(bad, dtype=torch.int64, device=self.dev)])
        if self.dist is not None:
            mx = t.clone()
            self.dist.all_reduce(mx, op=self.dist.ReduceOp.MAX)
            sm = t[2:].clone()
            self.dist.all_reduce(sm, op=self.dist.ReduceOp.SUM)
            return int(sm[0].item()), bool(mx[0].item() == -mx[1].item())
        return bad, True

    def gather_bytes_per_step(self):
        """bytes every GPU RECEIVES per step through the collective"""
        if self.gather == "full":
            return self.world * self.B * (self.cap * SLOT_BYTES + 4)
        return self.world * self.B * 8 if self.gather == "counts" else 0

    def match_ms(self):
        return sum(e0.elapsed_time(e1) for e0, e1 in self.match_events) / max(1, len(self.match_events))


def host_io_rate(ex, frames, slot_frames, rounds, pinned):
    """frames/s through the pipelined ring with HOST pointers in and out (orbfe_stream_*): `frames` ([n][H][W] u8 in
    host memory, pinned or pageable) are submitted slot by slot, the ring is kept full, and every collected slot is
    copied into caller arrays (keypoints, descriptors, counts) like orbfe_extract_batch does."""
    src = torch.from_numpy(frames).pin_memory().numpy() if pinned else frames
    st = ex.stream(slots=3, slot_frames=slot_frames)
    n_total = len(src)
    chunks = [(i, min(slot_frames, n_total - i)) for i in range(0, n_total, slot_frames)]

    def run(n_rounds):
        done = 0
        todo = [c for _ in range(n_rounds) for c in chunks]
        pos = 0
        while pos < len(todo) or st.in_flight():
            while pos < len(todo):
                lo, n = todo[pos]
                if not st.submit(src[lo:lo + n]):
                    break
                pos += 1
            done += st.collect_raw()[0]
        return done

    run(3)  # warm: every slot of the ring used at least once (its pinned blocks touched and mapped), kernels loaded
    best = 0.0
    for _ in range(2):  # steady state: the faster of two timed repetitions (the first one on a fresh ring / fresh result arrays
        t0 = time.perf_counter()  # reads up to 25 % low on some boxes: tools/host_io_order.py)
        done = run(rounds)
        best = max(best, done / (time.perf_counter() - t0))
    st.close()
    return best


def host_io_match_rate(ex, frames, slot_frames, rounds):
    """frames/s of extract + isInFrustum + SearchByProjection with HOST pointers in and out through the ring
    (orbfe_stream_submit_track / collect_track): pinned frames by pointer, every frame's N_MAP_POINTS local map points by id
    out of a map resident in HBM (orbfe_map_*: world position + descriptor uploaded once, as a SLAM map would be), the
    frame's own pose (one frustum block per frame).  The map points are the C3 recipe back-projected through the pose, so
    the matcher sees the same projections / levels / descriptors as in the headline measurement."""
    import orbfe
    W, H = ex.W, ex.H
    n_total, M = len(frames), N_MAP_POINTS
    res = []
    for i in range(0, n_total, ex.max_batch):
        res += ex.extract_batch(list(frames[i:i + ex.max_batch]))
    F = orbfe.Frustum()
    F.rcw[0] = F.rcw[4] = F.rcw[8] = 1.0
    F.min_x, F.max_x, F.min_y, F.max_y = 0.0, float(W), 0.0, float(H)
    F.fx = F.fy = 458.0
    F.cx, F.cy, F.mbf = 0.5 * W, 0.5 * H, 40.0
    F.log_scale_factor, F.n_levels, F.camera_model = float(np.log(np.float32(1.2))), ex.nlevels, 0
    rng = np.random.default_rng(99)
    pts = np.zeros((n_total, M), orbfe.WP_DTYPE)
    mpd = np.zeros((n_total, M, 32), np.uint8)
    for b, (kp, desc, _) in enumerate(res):
        mps, mpd[b] = make_map_points(kp, len(kp), desc, M, rng, ex.nlevels, orbfe.MP_DTYPE)
        z = rng.uniform(2.0, 8.0, M)
        x, y = (mps["proj_x"] - F.cx) / F.fx * z, (mps["proj_y"] - F.cy) / F.fy * z
        pts[b]["x"], pts[b]["y"], pts[b]["z"] = x, y, z
        d = np.sqrt(x * x + y * y + z * z)
        pts[b]["max_distance"] = d * 1.2 ** (mps["level"] - 0.5)  # PredictScale: ceil(log(max / d) / log 1.2) == level
        pts[b]["min_distance"] = pts[b]["max_distance"] / 1.2 ** (ex.nlevels - 1)
        pts[b]["observations"] = mps["observations"]
    mp = orbfe.MapPoints(ex, n_total * M)
    mp.update(np.arange(n_total * M), pts.reshape(-1), mpd.reshape(-1, 32))
    ids = np.arange(n_total * M, dtype=np.int32).reshape(n_total, M)
    src = torch.from_numpy(frames).pin_memory().numpy()
    frusta = (orbfe.Frustum * slot_frames)(*([F] * slot_frames))
    st = ex.stream(slots=3, slot_frames=slot_frames)
    st.enable_track(mp, M, GRID[0], GRID[1], 0.0, 0.0, float(W), float(H))
    chunks = [(i, min(slot_frames, n_total - i)) for i in range(0, n_total, slot_frames)]
    stats = {"matches": 0, "frames": 0}

    def run(n_rounds):
        done = 0
        todo = [c for _ in range(n_rounds) for c in chunks]
        pos = 0
        while pos < len(todo) or st.in_flight():
            while pos < len(todo):
                lo, n = todo[pos]
                if not st.submit_track(src[lo:lo + n], frusta if n == slot_frames else (orbfe.Frustum * n)(*([F] * n)), ids[lo:lo + n],
                                       MATCH_TH, MATCH_NN):
                    break
                pos += 1
            nf, _kp, _desc, _n, _per, _match, nm = st.collect_track_raw()
            stats["matches"] += int(nm[:nf].sum())
            stats["frames"] += nf
            done += nf
        return done

    run(3)  # warm: every slot of the ring used at least once
    best = 0.0
    for _ in range(2):  # the faster of two timed repetitions, as host_io_rate
        t0 = time.perf_counter()
        done = run(rounds)
        best = max(best, done / (time.perf_counter() - t0))
    st.close()
    mp.close()
    return best, stats["matches"] / max(1, stats["frames"])


_RESULT_FD = None


def claim_stdout():
    """From here on file descriptor 1 of this process is the RESULT channel only: libraries that print to the C-level stdout (RCCL
    writes its version banner there when the first communicator is made) are sent to stderr, the JSON line goes out through a
    duplicate of the original descriptor (emit_line).  The driver reads ONE JSON line from stdout."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit_line(text):
    sys.stdout.flush()
    if _RESULT_FD is None:
        print(text, flush=True)
    else:
        os.write(_RESULT_FD, (text + "\n").encode())


def _pct(v, q):
    return float(np.percentile(np.asarray(v, np.float64), q))


def sustained_run(runner, ex, dev, dist, frames_total, seconds, rank, world, dump=None):
    """The same step loop as the headline region, for >= `seconds` of wall time: one HIP event behind every step (its
    duration = the distance to the previous step's event), and every ~1/64th of the run one clock probe
    (orbfe_debug_clock_probe, 20 us, one wave) on a stream of its own released by that step's event, so the probes are
    spread over the region and read the clock the extraction kernels run at.  Wall time between barriers, MAX over ranks, like
    the headline.  The reference's cadence is a continuous stream (mono_inertial_node.cpp:207), not a burst."""
    # the headline's few steps are a poor estimate of the step time: calibrate on 64 untimed steps (MAX over ranks, so that every
    # rank derives the same step count -- the per-step collective needs that)
    torch.cuda.synchronize(dev)
    tc = time.perf_counter()
    for _ in range(64):
        runner.step()
    torch.cuda.synchronize(dev)
    est_step_s = max((time.perf_counter() - tc) / 64, 1e-6)
    if dist is not None:
        t = torch.tensor([est_step_s], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        est_step_s = float(t[0].item())
    n_first = max(200, int(np.ceil(seconds * 1.04 / est_step_s)))
    every = max(1, n_first // 64)
    probes = torch.zeros((4096, 2), dtype=torch.int64, device=dev)
    s3 = torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)
    ex.clock_probe(probes[0].data_ptr(), 20, s3.cuda_stream)  # the clock of the idle chip, for comparison
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    evs = [torch.cuda.Event(enable_timing=True)]
    evs[0].record(runner.s2)
    k = 1

    def enqueue(count):
        nonlocal k
        for _ in range(count):
            i = len(evs) - 1
            evs.append(torch.cuda.Event(enable_timing=True))
            runner.step(done_event=evs[-1])
            if i % every == every // 2 and k < len(probes):
                s3.wait_event(evs[-1])
                ex.clock_probe(probes[k].data_ptr(), 20, s3.cuda_stream)
                k += 1

    enqueue(n_first)
    # single process: look at the events 32 steps before the end of what is enqueued (the queue never drains) and top up until the
    # region covers `seconds` (with several ranks the count is fixed by the calibration above: agreeing on a top-up would drain it)
    for _ in range(64 if dist is None else 0):
        j = max(1, len(evs) - 1 - 32)
        evs[j].synchronize()
        per_step = evs[0].elapsed_time(evs[j]) / j
        more = int(np.ceil((seconds * 1e3 * 1.03 - per_step * (len(evs) - 1)) / max(per_step, 1e-3)))
        if more <= 0 or len(evs) > 400000:
            break
        enqueue(max(more, 16))
    n_steps = len(evs) - 1
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    dt_local = dt = time.perf_counter() - t0
    d_ms = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(n_steps)], np.float64)
    p50 = _pct(d_ms, 50)
    per_rank = [p50]
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0].item())
        g = torch.zeros(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(g, torch.tensor([p50], dtype=torch.float64, device=dev))
        per_rank = [float(x) for x in g.tolist()]
    pr = probes[:k].cpu().numpy().astype(np.float64)
    mhz = pr[:, 0] / np.maximum(pr[:, 1], 1.0) * 100.0  # shader cycles per 100 MHz tick (MI355X_MICROARCH.md, DVFS item 6)
    run = mhz[1:]
    hundred = min(100, n_steps // 2)
    # a step's event sits behind its MATCHER on the second stream: when one step's matcher slips into the next extraction the two
    # events land late / on time -- a long step followed by a short one, the pair summing to two medians (no throughput lost).
    # The rolling mean over 8 consecutive steps separates that completion jitter from steps that really took longer.
    tail_ms = d_ms[1:] if n_steps > 9 else d_ms
    win = np.convolve(tail_ms, np.ones(8) / 8.0, mode="valid") if len(tail_ms) >= 8 else tail_ms
    if dump and rank == 0:
        with open(dump, "w") as f:
            json.dump({"ms_per_step": [round(float(x), 5) for x in d_ms], "sclk_mhz": [round(float(x), 1) for x in mhz], "probe_every_steps": every}, f)
    return {"seconds": dt, "steps": n_steps, "value": frames_total * n_steps / dt, "unit": "frames/s",
            "ms_per_step": {"p50": p50, "p99": _pct(d_ms, 99), "max": float(d_ms.max()), "min": float(d_ms.min()),
                            "mean": float(d_ms.mean()), "wall_over_steps": dt_local / n_steps * 1e3,
                            # the first step fills the two-stream pipeline (its extraction and its matcher run back to back)
                            "max_after_first_step": float(d_ms[1:].max()) if n_steps > 1 else float(d_ms.max()),
                            "window8": {"p50": _pct(win, 50), "p99": _pct(win, 99), "max": float(win.max()),
                                        "what": "rolling mean of 8 consecutive steps (from the second step on): a matcher that finishes late "
                                                "makes ITS step long and the next one short by the same amount; the window cancels that"}},
            "first_100_steps_value": frames_total * hundred / (d_ms[:hundred].sum() * 1e-3),
            "last_100_steps_value": frames_total * hundred / (d_ms[-hundred:].sum() * 1e-3),
            "per_rank_p50_ms": per_rank,
            "sclk_mhz": {"start": float(np.median(run[:3])) if len(run) else None, "end": float(np.median(run[-3:])) if len(run) else None,
                         "min": float(run.min()) if len(run) else None, "max": float(run.max()) if len(run) else None,
                         "mean": float(run.mean()) if len(run) else None, "idle_before": float(mhz[0]), "probes": int(len(run)),
                         "method": "in-kernel: delta s_memtime / delta s_memrealtime x 100 MHz over 20 us, one wave on its own stream "
                                   "beside the workload (rank 0's GPU)"},
            "what": "step durations and first/last-100 figures are rank %d's (event to event on the stream the step ends on); "
                    "`value` = all ranks' frames / MAX-over-ranks wall time of the region" % rank}


def verify_last_step(runner, frames_host, state, B, cap, M, cfg, n_frames=8):
    """The headline says bit-exact: take `n_frames` frames spread over the batch of the LAST step that ran (whatever timed region
    that was), run the CPU oracle on the same frames and the same map points, and compare the device buffers byte for byte.
    The oracle is the checker here, outside every timed region (src/ORBextractor.cc:543-585, src/ORBmatcher.cc:31-123)."""
    import oracle_py as O
    import orbfe
    b, fs = runner.last
    W, H = cfg[6], cfg[7]
    idx = sorted(set(int(round(x)) for x in np.linspace(0, B - 1, min(n_frames, B))))
    sel = torch.tensor(idx, device=b["n"].device)
    n_h = b["n"][sel].cpu().numpy()
    kp_h = b["kp"][sel].cpu().numpy().reshape(len(idx), cap * 24).view(orbfe.KP_DTYPE).reshape(len(idx), cap)
    desc_h = b["desc"][sel].cpu().numpy()
    ref = O.Extractor(*cfg)
    kp_ok = match_ok = True
    first_bad = None
    if M:
        match_h = b["match"][sel].cpu().numpy()
        nm_h = b["nmatch"][sel].cpu().numpy()
        mps_h = state["mps"][fs].cpu().numpy().view(orbfe.MP_DTYPE).reshape(-1, M)
        mpd_h = state["mpd"][fs].cpu().numpy().reshape(-1, M, 32)
    for j, i in enumerate(idx):
        kp_r, desc_r, _ = ref.extract(frames_host[fs][i])
        n = int(n_h[j])
        same = n == len(kp_r) and kp_h[j, :n].tobytes() == kp_r.tobytes() and np.array_equal(desc_h[j, :n], desc_r)
        kp_ok &= bool(same)
        if M:
            if len(kp_r):
                fv = O.make_frame_view(kp_r, desc_r, GRID[0], GRID[1], 0.0, 0.0, float(W), float(H), ref.scaleFactors)
                nm_r, match_r = O.search_by_projection(fv, mps_h[i].view(O.MP_DTYPE), mpd_h[i], None, MATCH_TH, MATCH_NN)
            else:
                nm_r, match_r = 0, np.zeros(0, np.int32)
            msame = same and int(nm_h[j]) == nm_r and np.array_equal(match_h[j, :n], match_r)
            match_ok &= bool(msame)
            same = same and msame
        if not same and first_bad is None:
            first_bad = i
    out = {"frames": len(idx), "frame_indices": idx, "frame_set": fs, "kp_desc_equal": kp_ok,
           "checker": "CPU oracle (oracle/, single thread) on the same frames%s, after the timed regions" % (" and map points" if M else "")}
    if M:
        out["match_equal"] = match_ok
    if first_bad is not None:
        out["first_mismatch_frame"] = first_bad
    return out


def launch_ranks(n, argv):
    """`bench.py --gpus N` started as ONE process: start N ranks (one per GPU) through torch.distributed.run on
    127.0.0.1 and relay rank 0's JSON line.  The parent never initialises the GPU (no torch.cuda call, no HIP call): the
    ranks are fresh child processes.  torchrun picks the rendezvous port itself (--standalone: c10d store on port 0), so
    two launches started together cannot race for one.  Returns the exit code of the launch."""
    import collections
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL (the host driver supports nothing else)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node=%d" % n, os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, text=True)
    line = None
    tail = collections.deque(maxlen=30)  # what the ranks said last, for the error message of a failed launch
    for out in proc.stdout:
        t = out.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        else:
            tail.append(out)
            sys.stderr.write(out)  # anything else the ranks print is not the result line
    rc = proc.wait()
    if rc != 0 or line is None:
        sys.stderr.write("bench.py: the %d-rank launch %s; last output of the ranks:\n%s" % (
            n, "failed with exit code %d" % rc if rc != 0 else "printed no result line", "".join(tail)))
        return rc if rc != 0 else 3
    print(line, flush=True)
    return 0


def load_image_dir(path, W, H, count):
    """--images DIR: grey frames of the workload geometry from image files (sorted by name; PGM / PNG / JPEG / BMP / TIFF
    via PIL).  A larger image is centre-cropped when it is at most 1.25x the geometry (EuRoC 752x480, TUM-VI 512x512 /
    1024x1024 frames fit as they are), otherwise resized with its aspect ratio kept and then centre-cropped; a smaller
    one is resized up.  The list is repeated to `count` frames.  Returns ([count][H][W] u8, number of files read)."""
    from PIL import Image
    names = sorted(f for f in os.listdir(path) if f.lower().endswith((".pgm", ".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff")))
    if not names:
        raise SystemExit("--images %s: no image files" % path)
    frames = []
    for f in names[:count]:
        im = Image.open(os.path.join(path, f)).convert("L")
        w, h = im.size
        if not (W <= w <= 1.25 * W and H <= h <= 1.25 * H):
            k = max(W / w, H / h)
            im = im.resize((max(W, int(round(w * k))), max(H, int(round(h * k)))), Image.BILINEAR)
            w, h = im.size
        x0, y0 = (w - W) // 2, (h - H) // 2
        frames.append(np.asarray(im.crop((x0, y0, x0 + W, y0 + H)), np.uint8))
    n_files = len(frames)
    while len(frames) < count:
        frames.append(frames[len(frames) % n_files])
    return np.stack(frames), n_files


def _median_ms(fn, reps, warm=5):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return float(np.median(ts)) * 1e3


def _dist_ms(fn, reps, warm=20):
    """p50 / p99 / max (and mean) of `reps` calls in ms: what a 20 Hz real-time consumer budgets against"""
    for _ in range(warm):
        fn()
    ts = np.empty(reps, np.float64)
    for i in range(reps):
        t = time.perf_counter()
        fn()
        ts[i] = time.perf_counter() - t
    ts *= 1e3
    return {"p50": _pct(ts, 50), "p99": _pct(ts, 99), "p999": _pct(ts, 99.9), "max": float(ts.max()), "mean": float(ts.mean()), "calls": int(reps)}


class MappingLoad:
    """The reference runs LocalMapping::Run on its own thread beside the tracking thread (src/System.cc; src/LocalMapping.cc:66-110):
    this context runs the mapping thread's calls in a loop on ITS OWN handle -- a new key frame uploaded (orbfe_keyframe_create),
    SearchForTriangulation against K = 20 resident neighbours in one launch (orbfe_match_triangulation_batch), a Fuse search
    against a key frame (orbfe_fuse_search, 2000 map points) and ComputeDistinctiveDescriptors for 200 map points -- while the
    caller measures the tracking thread's latency.  ctypes releases the GIL inside every call."""

    def __init__(self, cfg, device_index, frame):
        import threading
        import orbfe
        import oracle_py as O
        import frustum_scenarios as FS
        import test_distinct
        import test_fuse
        import test_triangulation_batch as TB
        from test_frustum import ON, PN
        self.ex = orbfe.ORBextractor(*cfg, device=device_index, max_batch=1)
        self.m = orbfe.ORBmatcher(self.ex)
        ex = self.ex
        kp, desc = ex.extractFeatures(frame)
        kpo = kp.view(O.KP_DTYPE)
        W, H = cfg[6], cfg[7]
        node1 = TB.nodes_of(kpo)
        K = 20
        nbs = [TB.neighbour(kpo, desc, 500 + k, True, False) for k in range(K)]
        self.kf2 = [orbfe.KeyFrame(ex, nb["kp"].view(orbfe.KP_DTYPE), nb["desc"], nb["node"], ex.mvScaleFactor) for nb in nbs]
        self.prm = [orbfe.tri_params(nb["F12"], nb["ep"], False, False, True) for nb in nbs]
        self.has2 = [nb["has"] for nb in nbs]
        self.has1 = (np.random.default_rng(9).random(len(kp)) < 0.3).astype(np.uint8)
        Fo, self.Fp = O.Frustum(), orbfe.Frustum()
        v = FS.fill_frustum(Fo, ON, W=float(W), H=float(H), n_levels=ex.nlevels, scale=cfg[2], seed=3)
        FS.fill_frustum(self.Fp, PN, W=float(W), H=float(H), n_levels=ex.nlevels, scale=cfg[2], seed=3)
        pts, self.fmpd, _, self.inv_s2 = test_fuse.scenario(kpo, desc, ex.mvScaleFactor, v, N_MAP_POINTS, 1, False)
        self.pts = pts.view(orbfe.WP_DTYPE)
        self.fv = orbfe.make_frame_view(kp, desc, GRID[0], GRID[1], 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
        self.doff, self.ddesc = test_distinct.make_sets(3, [int(x) for x in np.random.default_rng(0).integers(2, 20, 200)])
        self.kp, self.desc, self.node1 = kp, desc, node1
        self.stop, self.rounds, self.err = threading.Event(), 0, None
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _round(self):
        import orbfe
        ex = self.ex
        kf1 = orbfe.KeyFrame(ex, self.kp, self.desc, self.node1, ex.mvScaleFactor)
        orbfe.SearchForTriangulation_batch(ex, kf1, self.has1, self.kf2, self.has2, self.prm)
        kf1.close()
        self.m.Fuse_search(self.fv, self.inv_s2, None, self.Fp, 3.0, self.pts, self.fmpd)
        self.m.ComputeDistinctiveDescriptors(self.doff, self.ddesc)

    def _run(self):
        try:
            while not self.stop.is_set():
                self._round()
                self.rounds += 1
        except Exception as e:  # noqa: BLE001 -- reported by the caller
            self.err = e

    def __enter__(self):
        self._round()  # arenas grown, kernels loaded before the measurement starts
        self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        self.thread.join(timeout=60)
        for k in self.kf2:
            k.close()
        self.ex.close()
        if self.err is not None and exc[0] is None:
            raise self.err


def latency_block(cfg, device_index, frames, reps=200, tail_calls=5000):
    """The reference's call shape: ONE host image per call (src/Frame.cc:178-189: ExtractORB on the frame the node hands
    over at 20 Hz, ros2_ws/src/mono-inertial/src/mono_inertial_node.cpp:207), then one SearchByProjection
    (src/Tracking.cc:1115), and the node-side preparation + extraction chain (image_grabber.hpp:96-110).  Wall time per
    call through the C ABI (upload, kernels, download, synchronisation), median of `reps` calls, and the single-thread
    oracle's median on the same call."""
    import orbfe
    import oracle_py as O
    from orbfe import synth
    W, H = cfg[6], cfg[7]
    ex1 = orbfe.ORBextractor(*cfg, device=device_index, max_batch=1)
    m1 = orbfe.ORBmatcher(ex1)
    ref = O.Extractor(*cfg)
    pageable = [np.ascontiguousarray(f) for f in frames[:32]]
    pinned = [torch.from_numpy(f.copy()).pin_memory().numpy() for f in pageable]
    for f in pinned:  # the first DMA access to a fresh pinned allocation maps it
        ex1.extractFeatures(f)
    it = {"i": 0}

    def call_extract(src):
        def fn():
            it["i"] += 1
            return ex1.extractFeatures(src[it["i"] % len(src)])
        return fn

    out = {"unit": "ms per call, median of %d" % reps, "calls": reps}
    out["extract_pageable_ms"] = _median_ms(call_extract(pageable), reps)
    out["extract_pinned_ms"] = _median_ms(call_extract(pinned), reps)
    out["extract_oracle_ms"] = _median_ms(lambda: ref.extract(pageable[0]), 15, 2)
    kp, desc = ex1.extractFeatures(pageable[0])
    rng = np.random.default_rng(11)
    mps, mpd = make_map_points(kp, len(kp), desc, N_MAP_POINTS, rng, ex1.nlevels, orbfe.MP_DTYPE)
    fv = orbfe.make_frame_view(kp, desc, GRID[0], GRID[1], 0.0, 0.0, float(W), float(H), ex1.mvScaleFactor)
    out["match_projection_ms"] = _median_ms(lambda: m1.SearchByProjection(fv, mps, mpd, MATCH_TH, False, 0.0, MATCH_NN, None), reps)
    kp_r, desc_r, _ = ref.extract(pageable[0])
    fvo = O.make_frame_view(kp_r, desc_r, GRID[0], GRID[1], 0.0, 0.0, float(W), float(H), ref.scaleFactors)
    out["match_projection_oracle_ms"] = _median_ms(lambda: O.search_by_projection(fvo, mps.view(O.MP_DTYPE), mpd, None, MATCH_TH, MATCH_NN), 15, 2)
    out["match_projection_map_points"] = N_MAP_POINTS
    # the whole per-frame chain of the tracking thread (src/Tracking.cc:152-173,1059-1115): ExtractORB -> isInFrustum over the
    # local map -> SearchByProjection, as three host-synchronous calls and as ONE submission (orbfe_track_frame: one graph)
    import frustum_scenarios as FS
    Fp, Fo = orbfe.Frustum(), O.Frustum()
    snake = {k: k for k in ("rcw", "tcw", "twc", "min_x", "max_x", "min_y", "max_y", "fx", "fy", "cx", "cy", "k1", "k2", "k3", "k4",
                            "mbf", "log_scale_factor", "n_levels", "camera_model")}
    camel = dict(snake, min_x="minX", max_x="maxX", min_y="minY", max_y="maxY", log_scale_factor="logScaleFactor", n_levels="nLevels",
                 camera_model="cameraModel")
    v = FS.fill_frustum(Fp, snake, W=float(W), H=float(H), n_levels=ex1.nlevels, scale=cfg[2], seed=21)
    FS.fill_frustum(Fo, camel, W=float(W), H=float(H), n_levels=ex1.nlevels, scale=cfg[2], seed=21)
    wpts, wdesc = FS.world_points_on_keypoints(kp, desc, v, N_MAP_POINTS, np.random.default_rng(12), ex1.nlevels, orbfe.WP_DTYPE)
    trk = orbfe.FrameTracker(ex1, GRID[0], GRID[1], 0.0, 0.0, float(W), float(H))
    out["project_map_points_ms"] = _median_ms(lambda: m1.isInFrustum_batch(Fp, wpts), reps)
    mps3, _ = m1.isInFrustum_batch(Fp, wpts)

    def three_calls():
        it["i"] += 1
        k, d = ex1.extractFeatures(pinned[it["i"] % len(pinned)])
        mp, _ = m1.isInFrustum_batch(Fp, wpts)
        f = orbfe.make_frame_view(k, d, GRID[0], GRID[1], 0.0, 0.0, float(W), float(H), ex1.mvScaleFactor)
        return m1.SearchByProjection(f, mp, wdesc, MATCH_TH, False, 0.0, MATCH_NN, None)

    def fused(src):
        def fn():
            it["i"] += 1
            return trk.TrackFrame(src[it["i"] % len(src)], Fp, wpts, wdesc, MATCH_TH, MATCH_NN)
        return fn

    out["three_calls_ms"] = _median_ms(three_calls, reps)
    out["track_frame_ms"] = _median_ms(fused(pinned), reps)
    out["track_frame_pageable_ms"] = _median_ms(fused(pageable), reps)
    got = trk.TrackFrame(pageable[0], Fp, wpts, wdesc, MATCH_TH, MATCH_NN)
    out["track_frame_matches"] = int(got["nmatches"])
    # the same call with the local map points named by id out of a map resident in HBM (8 KB of ids instead of 128 KB of points)
    mp_res = orbfe.MapPoints(ex1, N_MAP_POINTS)
    mp_res.update(np.arange(N_MAP_POINTS), wpts, wdesc)
    ids_res = np.arange(N_MAP_POINTS, dtype=np.int32)
    ids_res = np.where(wpts["skip"] != 0, ~ids_res, ids_res).astype(np.int32)  # "mnLastFrameSeen == this frame" travels with the id

    def fused_map():
        it["i"] += 1
        return trk.TrackFrameMap(pinned[it["i"] % len(pinned)], Fp, mp_res, ids_res, MATCH_TH, MATCH_NN)

    out["track_frame_map_ms"] = _median_ms(fused_map, reps)
    got_map = trk.TrackFrameMap(pageable[0], Fp, mp_res, ids_res, MATCH_TH, MATCH_NN)
    out["track_frame_map_equals_track_frame"] = bool(got_map["nmatches"] == got["nmatches"] and np.array_equal(got_map["match"], got["match"]))

    def oracle_chain():
        k, d, _ = ref.extract(pageable[0])
        mp, _ = O.is_in_frustum(Fo, wpts.view(O.WP_DTYPE))
        f = O.make_frame_view(k, d, GRID[0], GRID[1], 0.0, 0.0, float(W), float(H), ref.scaleFactors)
        return O.search_by_projection(f, mp, wdesc, None, MATCH_TH, MATCH_NN)

    n_o, match_o = oracle_chain()
    out["track_frame_equals_oracle"] = bool(n_o == got["nmatches"] and np.array_equal(match_o, got["match"]))
    out["track_frame_oracle_ms"] = _median_ms(oracle_chain, 10, 1)
    mp_res.close()
    # the third per-frame chain of the tracking thread, while the map is being initialised (src/Tracking.cc:566-607): ExtractORB ->
    # SearchForInitialization(mInitialFrame, mCurrentFrame, 40, 0.45, true) against an initial frame resident in HBM, ONE submission
    ini = orbfe.InitialFrame(ex1, kp, desc)

    def ini_chain():
        it["i"] += 1
        return trk.TrackInitialization(pinned[it["i"] % len(pinned)], ini, 40, 0.45, True)

    out["track_initialization_ms"] = _median_ms(ini_chain, reps)
    got_i = trk.TrackInitialization(pageable[1], ini, 40, 0.45, True)
    kp_o1, desc_o1, _ = ref.extract(pageable[1])
    n_i, m12_i = O.search_for_initialization(fvo, O.make_frame_view(kp_o1, desc_o1, GRID[0], GRID[1], 0.0, 0.0, float(W), float(H), ref.scaleFactors),
                                             40, 0.45, True)
    out["track_initialization_matches"] = int(got_i["nmatches"])
    out["track_initialization_equals_oracle"] = bool(got_i["kp"].tobytes() == kp_o1.tobytes() and n_i == got_i["nmatches"] and
                                                     np.array_equal(m12_i, got_i["matches12"]))
    ini.close()
    if tail_calls > 0:
        # ---- tail latency of the two per-frame chains (src/Tracking.cc:152-173,925-930 / :825-835), alone and beside the mapping
        #      thread (src/LocalMapping.cc:66-110) on its own handle ----
        import vocab_synth as vs
        tvoc = vs.spread_first_level(vs.make_tree(10, 6, seed=17, early_leaf_p=0.02), 18)  # ORBvoc's shape: k = 10, L = 6
        voc = orbfe.ORBVocabulary(ex1, tvoc["childOff"], tvoc["childIdx"], tvoc["nodeDesc"], tvoc["wordId"], tvoc["weight"], 6)
        _, node_kf, w_kf = voc.transform(desc, 4)
        kf_res = orbfe.KeyFrame(ex1, kp, desc, np.where(w_kf > 0, node_kf, -1).astype(np.int32), ex1.mvScaleFactor)
        has_kf = (np.random.default_rng(4).random(len(kp)) < 0.8).astype(np.uint8)

        def ref_chain():
            it["i"] += 1
            return trk.TrackReferenceKeyFrame(pinned[it["i"] % len(pinned)], voc, 4, kf_res, has_kf, 0.75, True)

        out["track_reference_keyframe_matches"] = int(ref_chain()["nmatches"])
        tail = {"unit": "ms per call", "alone": {"track_frame": _dist_ms(fused(pinned), tail_calls),
                                                  "track_reference_keyframe": _dist_ms(ref_chain, tail_calls)}}
        with MappingLoad(cfg, device_index, pageable[0]) as load:
            r0, t0 = load.rounds, time.perf_counter()
            tail["loaded"] = {"track_frame": _dist_ms(fused(pinned), tail_calls),
                              "track_reference_keyframe": _dist_ms(ref_chain, tail_calls)}
            tail["mapping_rounds_per_s"] = (load.rounds - r0) / (time.perf_counter() - t0)
            # the lever for the tail: the tracking handle's stream at high priority (orbfe_set_stream_priority) -- its kernels are
            # dispatched ahead of the mapping thread's whenever both are queued (tools/tail_latency.py: profiles/r05_tail_latency.json)
            ex1.set_stream_priority(True)
            tail["loaded_high_priority"] = {"track_frame": _dist_ms(fused(pinned), tail_calls),
                                            "track_reference_keyframe": _dist_ms(ref_chain, tail_calls)}
        tail["what"] = ("%d calls each from pinned frames; `loaded` = the same calls while a second thread on its OWN handle loops over "
                        "the mapping thread's calls (orbfe_keyframe_create + orbfe_match_triangulation_batch K = 20 + orbfe_fuse_search "
                        "%d map points + orbfe_distinctive_descriptors 200 sets; mapping_rounds_per_s of them ran meanwhile -- a real "
                        "mapping thread issues a few such rounds per key frame, i.e. tens per second); `loaded_high_priority` = the same "
                        "with orbfe_set_stream_priority(tracking handle, 1)" % (tail_calls, N_MAP_POINTS))
        out["tail"] = tail
        out["track_frame_ms_p99"] = {k: tail[k]["track_frame"]["p99"] for k in ("alone", "loaded", "loaded_high_priority")}
        out["track_reference_keyframe_ms_p99"] = {k: tail[k]["track_reference_keyframe"]["p99"] for k in ("alone", "loaded", "loaded_high_priority")}
        kf_res.close()
        voc.close()
    # node-side chain at the node's own configuration (mono_inertial_node.cpp:20,59-71): 2048x1536 BGR -> 614x460 grey -> extract
    SW, SH, DW, DH = 2048, 1536, 614, 460
    pcfg = (cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], cfg[5], DW, DH)
    exp = orbfe.ORBextractor(*pcfg, device=device_index, max_batch=1)
    bgr = synth.colour_image(SW, SH, 1)
    map1, map2 = synth.fisheye_maps(SW, SH, 1)
    prep = orbfe.ImagePreparer(exp, map1, map2, DW, DH)
    bgr_pinned = torch.from_numpy(bgr.copy()).pin_memory().numpy()
    out["prepare_and_extract_pinned_ms"] = _median_ms(lambda: prep.extract(bgr_pinned), reps)
    out["prepare_and_extract_pageable_ms"] = _median_ms(lambda: prep.extract(bgr), max(20, reps // 4))
    refp = O.Extractor(*pcfg)

    def oracle_chain():
        refp.extract(O.prepare_image(bgr, map1, map2, DW, DH))
    out["prepare_and_extract_oracle_ms"] = _median_ms(oracle_chain, 3, 1)
    out["what"] = ("one %dx%d host frame per call through orbfe_extract (one captured hipGraph: H2D, kernels, D2H, sync); "
                   "orbfe_match_projection with %d map points, host pointers; three_calls = orbfe_extract + orbfe_project_map_points + "
                   "orbfe_match_projection back to back (three host synchronisations), track_frame = the same chain through "
                   "orbfe_track_frame (one graph, one synchronisation); track_initialization = orbfe_track_initialization (extract + "
                   "SearchForInitialization against a resident initial frame, one graph); orbfe_prepare_and_extract %dx%d BGR -> %dx%d; "
                   "`*_oracle_ms` = the single-thread C oracle on the same call" % (W, H, N_MAP_POINTS, SW, SH, DW, DH))
    prep.close()
    return out


def launcher_selftest(a, rank, world):
    """--selftest-launcher (tests/test_distributed_cpu.py): the rank bookkeeping of this script -- launch, process group,
    StepRunner buffer rotation, the per-step collective, the MAX-over-ranks timing and rank 0's JSON line -- on CPU tensors
    over gloo with a stub in place of the HIP calls.  It measures nothing: metric "selftest", value 0."""
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, cap, nl = 3, 5, 2

    def extract_fn(b, fs):
        for j in range(B):
            b["n"][j] = rank * B + j + 1

    def match_fn(b, fs):
        for j in range(B):
            b["nmatch"][j] = rank + j

    mode = "none" if a.no_gather else (a.gather if a.gather != "auto" else "counts")
    r = StepRunner(torch.device("cpu"), B, cap, nl, extract_fn, match_fn, dist, world, mode, True, 1)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        r.step()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if dist.get_world_size() != a.gpus:
        raise SystemExit("selftest: --gpus %d but the process group has %d ranks" % (a.gpus, dist.get_world_size()))
    gv = r.verify_gather()  # this rank's slots of the gathered arrays == what it packed; the same bytes on every rank
    if rank == 0:
        counts = r.g_cnt.view(world, B, 2).tolist() if mode == "counts" else None
        emit_line(json.dumps({"metric": "selftest", "value": 0.0, "unit": "none", "n_gpus": dist.get_world_size(), "steps": a.steps,
                          "gather_verified": None if gv is None else {"mismatching_slots_all_ranks": gv[0], "identical_on_all_ranks": gv[1],
                                                                       "ok": gv[0] == 0 and gv[1]},
                          "warmup": a.warmup, "data": "stub", "config": {"workload": "launcher self-test (no device work)",
                                                                          "gather": mode, "gather_bytes_per_step": r.gather_bytes_per_step()},
                          "gathered_counts": counts}))
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="frames per step per GPU in weak-scaling mode")
    ap.add_argument("--workload", default="euroc_752x480", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="auto: strong (512 frames in total, BASELINE config 4) for batched_1280x720, weak otherwise")
    ap.add_argument("--frame-sets", type=int, default=3,
                    help="distinct synthetic frame sets the steps rotate through (3 x 512 x 752x480 = 555 MB: more than "
                         "the 256 MB Infinity Cache, so level 0 is read from HBM)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads of the all-cores CPU baseline (0/1 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-io", action="store_true", help="skip the host-pointer ring measurement (value_host_io)")
    ap.add_argument("--texture-sweep", action="store_true", help="also time a low-texture and a 1/f-noise stream")
    ap.add_argument("--gather", default="auto", choices=["auto", "full", "counts", "none"],
                    help="per-step RCCL collective at N > 1.  auto: `full` (all_gather of the padded keypoints + descriptors + "
                         "match indices, BASELINE config 4) in strong mode, `counts` (8 B per frame) in weak mode")
    ap.add_argument("--no-gather", action="store_true", help="same as --gather none")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the per-step collective even at N=1 (a 1-rank process group): exercises the N>1 code path on one GPU")
    ap.add_argument("--emulate-world", type=int, default=0, metavar="K",
                    help="strong mode on ONE GPU: run rank 0's shard of a K-rank split (512 / K frames per step), no collective; "
                         "predicts the per-GPU step time of the K-GPU run")
    ap.add_argument("--images", default=None, metavar="DIR",
                    help="grey-convert the image files of DIR (sorted; PGM / PNG / JPEG via PIL, centre-cropped or resized to the "
                         "workload geometry) instead of the synthetic stream")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-call latency block")
    ap.add_argument("--latency-calls", type=int, default=200)
    ap.add_argument("--latency-tail-calls", type=int, default=5000,
                    help="calls behind the p50 / p99 / max of orbfe_track_frame and orbfe_track_reference_keyframe, alone and beside a "
                         "mapping thread on its own handle (0 = skip)")
    ap.add_argument("--sustained-seconds", type=float, default=5.0,
                    help="length of the second timed region (`sustained`: per-step p50 / p99 / max, measured clock); 0 = skip")
    ap.add_argument("--sustained-dump", default=None, metavar="FILE",
                    help="write every step duration of the sustained region (ms, rank 0) and the clock probes to FILE as JSON")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the oracle check of 8 frames of the last timed step (`verified`) and the gather check")
    ap.add_argument("--selftest-launcher", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--split", type=int, default=1,
                    help="extract the batch as this many sub-batches, each on its own handle and HIP stream, staggered by the order of "
                         "their launches (the FAST kernel of one sub-batch beside the latency-bound kernels of the next)")
    ap.add_argument("--no-match", action="store_true", help="extract only")
    ap.add_argument("--no-overlap", action="store_true", help="match on the extraction stream (no 2-stream pipelining)")
    ap.add_argument("--lib", default=None, help="load this build of liborbfe instead (tools/: the timing-only ablation build)")
    a = ap.parse_args()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as one process: become the launcher of N ranks BEFORE anything touches the GPU
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))

    # both launch forms (ours above, the driver's torch.distributed.run) run the ranks with dmabuf IPC: set before the
    # first torch.cuda / HIP call of this process
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    claim_stdout()  # stdout carries the one result line; whatever else a library prints there goes to stderr
    if a.selftest_launcher:
        return launcher_selftest(a, rank, world)
    if a.emulate_world and (world > 1 or a.emulate_world < 1):
        raise SystemExit("--emulate-world K needs K >= 1 and a single-GPU run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # everything (our kernels, torch copies, the RCCL gather) is ordered on ONE explicit stream; the
    # default stream's handle is 0, which the C ABI reads as "use the handle's own stream"
    # BENCH_S1_PRIORITY: experiment knob (-1 = the extraction stream at high priority, so that the matcher's kernels on the second
    # stream are dispatched only into slots the extraction kernels leave free)
    torch.cuda.set_stream(torch.cuda.Stream(dev, priority=int(os.environ.get("BENCH_S1_PRIORITY", "0"))))
    dist = None
    if world > 1 or a.force_gather:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # --force-gather outside a launcher (one rank): a free port, not a fixed one
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if dist.get_world_size() != a.gpus:  # n_gpus of the result line is the group that ran, never the flag
            raise SystemExit("--gpus %d but the RCCL group has %d ranks" % (a.gpus, dist.get_world_size()))
        world = dist.get_world_size()

    import orbfe
    from orbfe import synth
    from orbfe.shard import shard_range
    if a.lib:
        orbfe.LIB_PATH = os.path.abspath(a.lib)

    cfg = WORKLOADS[a.workload]
    W, H = cfg[6], cfg[7]
    scaling = a.scaling if a.scaling != "auto" else ("strong" if a.workload == "batched_1280x720" else "weak")
    emu = a.emulate_world if scaling == "strong" else 0
    if a.emulate_world and not emu:
        raise SystemExit("--emulate-world applies to strong scaling (--workload batched_1280x720 or --scaling strong)")
    if scaling == "strong":
        lo, hi = shard_range(C4_TOTAL_FRAMES, rank, emu or world)
        B, frame0 = hi - lo, lo
        frames_total = B if emu else C4_TOTAL_FRAMES  # emulation: `value` counts the frames this one GPU really processed
        if B < 1:
            raise SystemExit("more ranks than frames")
    else:
        B, frame0, frames_total = a.batch, rank * a.batch, world * a.batch
    Bpad = (C4_TOTAL_FRAMES + world - 1) // world if (scaling == "strong" and not emu) else B  # gather slots per rank
    M = 0 if a.no_match else N_MAP_POINTS
    ex = orbfe.ORBextractor(*cfg, device=local_rank, max_batch=max(B, Bpad))
    matcher = orbfe.ORBmatcher(ex)
    cap = ex.cap
    n_sets = max(1, a.frame_sets)

    # ---- synthetic streams, resident in HBM before the timed region; the steps rotate through n_sets of them ----
    def gen_sets(stream_fn):
        sets = [np.stack(list(stream_fn(W, H, B, 1000 * s + frame0))) for s in range(n_sets)]
        return sets, [torch.from_numpy(f).to(dev) for f in sets]

    n_image_files = 0
    if a.images:
        imgs, n_image_files = load_image_dir(a.images, W, H, n_sets * B)
        frames_sets = [imgs[k * B:(k + 1) * B] for k in range(n_sets)]
        d_gray = [torch.from_numpy(np.ascontiguousarray(f)).to(dev) for f in frames_sets]
    else:
        frames_sets, d_gray = gen_sets(lambda w, h, n, i0: synth.stream(w, h, n, index0=i0))
    gather = "none"
    if (world > 1 or a.force_gather) and not a.no_gather and not emu:
        gather = a.gather if a.gather != "auto" else ("full" if scaling == "strong" else "counts")
    state = {"gray": d_gray, "mps": None, "mpd": None}

    n_split = max(1, min(a.split, B))
    sub = [(k * B // n_split, (k + 1) * B // n_split) for k in range(n_split)]
    sub_ex = [ex] if n_split == 1 else [orbfe.ORBextractor(*cfg, device=local_rank, max_batch=hi - lo) for lo, hi in sub]
    sub_streams = [torch.cuda.Stream(dev) for _ in sub] if n_split > 1 else []
    sub_events = [torch.cuda.Event() for _ in sub] if n_split > 1 else []
    fork_event = torch.cuda.Event() if n_split > 1 else None

    def extract_fn(b, fs):
        if n_split == 1:
            ex.extract_batch_device(state["gray"][fs].data_ptr(), W * H, W, B, b["kp"].data_ptr(), b["desc"].data_ptr(),
                                    b["n"].data_ptr(), b["per"].data_ptr(), runner.s1.cuda_stream)
            return
        fork_event.record(runner.s1)
        nl = ex.nlevels
        for (lo, hi), e, st, ev in zip(sub, sub_ex, sub_streams, sub_events):
            st.wait_event(fork_event)
            e.extract_batch_device(state["gray"][fs].data_ptr() + lo * W * H, W * H, W, hi - lo, b["kp"].data_ptr() + lo * cap * 24,
                                   b["desc"].data_ptr() + lo * cap * 32, b["n"].data_ptr() + lo * 4, b["per"].data_ptr() + lo * nl * 4,
                                   st.cuda_stream)
            ev.record(st)
        for ev in sub_events:
            runner.s1.wait_event(ev)

    def match_fn(b, fs):
        matcher.SearchByProjection_batch_device(B, b["kp"].data_ptr(), b["desc"].data_ptr(), b["n"].data_ptr(), cap,
                                                GRID[0], GRID[1], 0.0, 0.0, float(W), float(H), M, state["mps"][fs].data_ptr(),
                                                state["mpd"][fs].data_ptr(), None, MATCH_TH, MATCH_NN, b["match"].data_ptr(),
                                                b["nmatch"].data_ptr(), stream=runner.s2.cuda_stream)

    runner = StepRunner(dev, Bpad, cap, ex.nlevels, extract_fn, match_fn if M else None, dist, world, gather,
                        not a.no_overlap, n_sets)

    def make_points(sets_gray):
        """map points (C3 recipe) from one untimed extraction per frame set; resident in HBM as well"""
        mps_d, mpd_d = [], []
        b0 = runner.bufs[0]
        state["gray"] = sets_gray
        for fs in range(n_sets):
            extract_fn(b0, fs)
            torch.cuda.synchronize(dev)
            kp_h = b0["kp"][:B].cpu().numpy().reshape(B, cap * 24).view(orbfe.KP_DTYPE).reshape(B, cap)
            desc_h = b0["desc"][:B].cpu().numpy()
            n_h = b0["n"][:B].cpu().numpy()
            rng = np.random.default_rng(1234 + rank + 77 * fs)
            mps_all = np.zeros((B, M), orbfe.MP_DTYPE)
            mpd_all = np.zeros((B, M, 32), np.uint8)
            for i in range(B):
                mps_all[i], mpd_all[i] = make_map_points(kp_h[i], int(n_h[i]), desc_h[i], M, rng, ex.nlevels, orbfe.MP_DTYPE)
            mps_d.append(torch.from_numpy(mps_all.view(np.uint8).reshape(-1)).to(dev))
            mpd_d.append(torch.from_numpy(mpd_all.reshape(-1)).to(dev))
        return mps_d, mpd_d

    if M:
        state["mps"], state["mpd"] = make_points(d_gray)

    def barrier():
        if dist is not None:
            dist.barrier()

    def timed_run(steps, warmup):
        for _ in range(warmup):
            runner.step()
        torch.cuda.synchronize(dev)
        ex.set_stage_timing(True)  # HIP events on the launch stream, inside the timed region
        runner.match_events = []
        barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            runner.step(timed=True)
        torch.cuda.synchronize(dev)
        barrier()
        dt = time.perf_counter() - t0
        stage_ms, ncalls = ex.stage_ms()
        ex.set_stage_timing(False)
        state["rank_dt"] = (dt, dt)
        if dist is not None:
            # MAX over ranks is the job's time; MIN beside it makes an imbalance between the GPUs visible in the one line
            t = torch.tensor([dt, -dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0].item())
            state["rank_dt"] = (-float(t[1].item()), dt)
        return dt, stage_ms, ncalls

    dt, stage_ms, ncalls = timed_run(a.steps, a.warmup)
    ex.device_status()  # device-side guard flags of the last chain (raises if any is set); outside the timed region

    # ---- the second timed region: the same loop for >= 5 s, per-step durations and the clock (the headline above is untouched) ----
    sustained = None
    if a.sustained_seconds > 0:
        sustained = sustained_run(runner, ex, dev, dist, frames_total, a.sustained_seconds, rank, world, a.sustained_dump)
        ex.device_status()
    # ---- the line checks itself: 8 frames of the last step that ran against the oracle; the gathered bytes against the packed ones ----
    verified, gather_verified, all_ok = None, None, True
    if not a.no_verify:
        torch.cuda.synchronize(dev)
        verified = verify_last_step(runner, frames_sets, state, B, cap, M, cfg)
        ok = verified["kp_desc_equal"] and verified.get("match_equal", True)
        gv = runner.verify_gather()
        if gv is not None:
            gather_verified = {"mismatching_slots_all_ranks": gv[0], "identical_on_all_ranks": gv[1], "ok": gv[0] == 0 and gv[1]}
            ok = ok and gather_verified["ok"]
        if dist is not None:  # every rank checks its own frames; one failing rank fails the job
            t = torch.tensor([0 if ok else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            verified["ranks_failing"] = int(t[0].item())
            ok = verified["ranks_failing"] == 0
        all_ok = ok

    b_last = runner.bufs[0]
    n_kp_mean = float(b_last["n"][:B].float().mean().item())
    n_match_mean = float(b_last["nmatch"][:B].float().mean().item()) if M else None
    if rank == 0:
        per_stage_bytes, b_frame = algorithmic_bytes_per_frame(ex.levelW, ex.levelH, n_kp_mean)
        stage_avg = {k: v / max(1, ncalls) for k, v in stage_ms.items()}
        if M:
            stage_avg["match_projection"] = runner.match_ms()
        dom = max(per_stage_bytes, key=lambda s: stage_avg[s])
        dom_ms = stage_avg[dom]
        dom_bytes = per_stage_bytes[dom] * B
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        total_ms = stage_avg["total"]
        what = "extract-only" if not M else "extract + SearchByProjection(%d map points/frame, grid %dx%d, th=%g, nnRatio=%g)" % (
            M, GRID[0], GRID[1], MATCH_TH, MATCH_NN)
        prof = profile_for(dom, a.workload, B)
        issue = None
        clk_hz = sustained["sclk_mhz"]["mean"] * 1e6 if sustained and sustained["sclk_mhz"]["mean"] else CLK_HZ
        if prof and prof.get("valu_per_wave") and prof.get("waves_per_launch"):
            wi = prof["waves_per_launch"] * prof["valu_per_wave"]
            cyc_w = prof.get("valu_cycles_per_instruction_weighted")  # tools/isa_mix.py: emitted ISA x profiles/r03_valu_rate.txt
            issue = {"valu_per_wave": prof["valu_per_wave"], "salu_per_wave": prof.get("salu_per_wave"),
                     "lds_per_wave": prof.get("lds_per_wave"), "waves": prof["waves_per_launch"],
                     "clock_mhz": clk_hz / 1e6,
                     "clock_source": "measured in this run (sustained.sclk_mhz.mean)" if clk_hz != CLK_HZ else "2400 MHz assumed (no probe)",
                     # waves x valu / (1024 SIMDs x clk / 2 x t): the guide's 2-cycle wave64 VALU rate, the floor of the issue time ...
                     "issue_frac": wi / (SIMDS * clk_hz / 2.0 * dom_ms * 1e-3),
                     # ... and with every opcode of the kernel's emitted ISA priced at the rate tools/valu_rate.hip measures for its
                     # group (2.1-2.6 cycles for the 16-bit min/max/add/sub and the 32-bit add/and/or/xor forms, 4.1-4.5 for packed,
                     # dot, permute and 32-bit min/max): the share of the kernel's duration its vector instructions need to issue
                     "valu_cycles_per_instruction_weighted": cyc_w,
                     "issue_frac_weighted": wi * cyc_w / (SIMDS * clk_hz * dom_ms * 1e-3) if cyc_w else None,
                     # counter-measured: SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) of the committed profile
                     "valu_busy": prof.get("valu_busy"),
                     "profile_clock_mhz": prof.get("effective_clock_mhz")}
        out = {
            "metric": "frames/sec ORB %s, %dx%d %d-level %d-feat; bit-exact kp/desc" % (
                "extract+match" if M else "extract", W, H, cfg[3], cfg[0]),
            "value": frames_total * a.steps / dt,
            "unit": "frames/s",
            "n_gpus": dist.get_world_size() if dist is not None else 1,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "ms_per_step_ranks": {"min": state["rank_dt"][0] / a.steps * 1e3, "max": state["rank_dt"][1] / a.steps * 1e3},
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic" if not a.images else "images: %d files of %s, grey, %dx%d" % (n_image_files, a.images, W, H),
            "config": {"workload": "%s %s, %s, %s, frames resident in HBM (%d frame sets in rotation), "
                                   "nFeatures=%d levels=%d scale=%.1f FAST %d/%d nFast=%d" % (
                                       a.workload, "image files" if a.images else "synthetic stream", what,
                                       "%d frames/step/GPU (weak scaling)" % B if scaling == "weak" else
                                       ("rank 0's shard of a %d-rank split of %d frames: %d frames per step on this ONE GPU (strong "
                                        "scaling emulated, BASELINE config 4)" % (emu, C4_TOTAL_FRAMES, B)) if emu else
                                       "%d frames per step IN TOTAL, %d on this rank (strong scaling, BASELINE config 4)" % (frames_total, B),
                                       n_sets, cfg[0], cfg[3], cfg[2], cfg[4], cfg[5], cfg[1]),
                       "frames_per_step": frames_total, "frames_per_step_per_gpu": B, "mean_keypoints_per_frame": n_kp_mean,
                       "mean_matches_per_frame": n_match_mean,
                       "gather": {"full": "rccl all_gather of the padded kp + desc + match slots per step (60 B per slot)",
                                  "counts": "rccl all_gather of the per-frame keypoint and match counts per step (8 B per frame); "
                                            "results stay on the GPU that made them",
                                  "none": "none"}[gather],
                       "gather_bytes_per_step": runner.gather_bytes_per_step(),
                       "streams": ("extract(step i+1) || match(step i), double-buffered outputs" if runner.nbuf == 2 else "single stream") +
                                  ("" if n_split == 1 else "; extraction in %d sub-batches on their own handles and streams" % n_split)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_peak": achieved / HBM_MEASURED_COPY_GBS,
                         "traffic": prof.get("hbm_bytes_per_launch") if prof else None,
                         "profiles_head": prof.get("git_head") if prof else None,
                         "profile_file": prof.get("file") if prof else None,
                         "issue": issue,
                         "limiter": "VALU issue, with the LDS byte gathers of the corner-score stage as the second limit: packed-16, dot, "
                                    "permute and 32-bit min/max instructions cost ~4.2 cycles per wave64 instruction per SIMD, the 16-bit "
                                    "min/max/sub and add/and/or/xor forms the FAST stages run on ~2.3 (profiles/r03_valu_rate.txt): "
                                    "`issue.issue_frac_weighted` of the kernel's time is vector issue, `issue.valu_busy` is the counter's "
                                    "reading of the same; 128 extra packed instructions per wave cost 81 % of their saturated price "
                                    "(profiles/r03_ab_experiments.json)",
                         "input_set_bytes": int(n_sets * B * W * H),
                         "kernel_ms_per_launch": dom_ms, "algorithmic_bytes_per_launch": dom_bytes,
                         "extract_pipeline_achieved_GBs": b_frame * B / (total_ms * 1e-3) / 1e9 if total_ms > 0 else 0.0,
                         "stage_ms_per_step": stage_avg},
        }
        if sustained is not None:
            out["sustained"] = sustained
            out["sustained"]["headline_over_sustained"] = out["value"] / sustained["value"]
        if verified is not None:
            out["verified"] = verified
        if gather_verified is not None:
            out["gather_verified"] = gather_verified
        if emu:
            out["emulate_world"] = {"world": emu, "frames_per_step_total": C4_TOTAL_FRAMES,
                                    "predicted_value_at_world": C4_TOTAL_FRAMES * a.steps / dt,
                                    "note": "this GPU ran 1/%d of BASELINE config 4 per step; with independent shards and the "
                                            "(unmeasured) all_gather hidden, %d GPUs finish the 512 frames in this step time" % (emu, emu)}
        if M:
            mm = stage_avg["match_projection"]
            out["matcher"] = {"ms_per_step": mm, "map_points_per_s": B * M / (mm * 1e-3),
                              # brute force is M x N 256-bit Hamming evaluations per frame; the grid-windowed
                              # search evaluates a small fraction of them, this is the rate a brute-force matcher would need
                              "pairs_per_s_bruteforce_equivalent": B * M * n_kp_mean / (mm * 1e-3),
                              "note": "stage time on its own stream while the next step's extraction runs beside it"}
        if not a.no_host_io and world == 1 and scaling == "weak":
            slot = min(256, B)
            try:
                # long enough (>= 0.1 s) that filling and draining the three-slot ring is a few per cent of the timed region
                out["value_host_io"] = host_io_rate(ex, frames_sets[0], slot, 40, True)
                out["value_host_io_pageable"] = host_io_rate(ex, frames_sets[0], slot, 20, False)
                mean_m = None
                if M:
                    out["value_host_io_match"], mean_m = host_io_match_rate(ex, frames_sets[0], slot, 20)
                out["host_io"] = {"unit": "frames/s", "what": "value_host_io / _pageable: extract only, host pointers in (pinned / pageable "
                                  "numpy rows) and out, ring of 3 slots x %d frames, H2D || kernels || D2H on three streams; "
                                  "value_host_io_match: the same ring with orbfe_stream_submit_track -- pinned frames by pointer, %d local "
                                  "map points per frame by id out of a map resident in HBM (orbfe_map_*), the frame's own pose: extract + "
                                  "isInFrustum + SearchByProjection, keypoints + descriptors + match indices back to host arrays" % (
                                      slot, N_MAP_POINTS),
                                  "mean_matches_per_frame_host_io_match": mean_m,
                                  # PCIe Gen5 x16, 63 GB/s per direction, full duplex: the upload (W*H bytes per frame) is the
                                  # larger direction (the ring downloads the padded result block: cap * 56 B per frame)
                                  "pcie_ceiling_frames_per_s": 63e9 / (W * H)}
            except orbfe.OrbfeError as e:  # report, do not lose the headline line
                out["host_io"] = {"error": str(e)}
        if a.texture_sweep and world == 1:
            sweep = {"default": out["value"]}
            for name, fn in (("low_texture", lambda w, h, n, i0: synth.lowtex_stream(w, h, n, index0=i0)),
                             ("pink_noise", lambda w, h, n, i0: synth.pink_stream(w, h, n, index0=i0))):
                _, dg = gen_sets(fn)
                state["gray"] = dg
                if M:
                    state["mps"], state["mpd"] = make_points(dg)
                nst = max(5, a.steps // 2)
                dts, st_ms, nc = timed_run(nst, 2)
                ex.device_status()
                sweep[name] = {"frames_per_s": frames_total * nst / dts,
                               "mean_keypoints_per_frame": float(runner.bufs[0]["n"][:B].float().mean().item()),
                               "fast_ms": st_ms["fast_nms_blur"] / max(1, nc)}
            out["texture_sweep"] = sweep
        if not a.no_latency and world == 1 and not emu:
            try:
                out["latency"] = latency_block(cfg, local_rank, frames_sets[0], a.latency_calls, a.latency_tail_calls)
            except orbfe.OrbfeError as e:
                out["latency"] = {"error": str(e)}
        if not a.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            sample = frames_sets[0][:64]
            fps, n = cpu_baseline(cfg, sample, a.cpu_seconds, bool(M), orbfe.MP_DTYPE, pin_core=0)
            out["cpu_baseline"] = {"value": fps, "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": "median frame time of %d frames (after 5 warm-up) of the same stream through the "
                                             "single-thread C oracle (%s) pinned to one core, host CPU %s, nproc=%d" % (
                                                 n, what.split("(")[0].strip(), host_cpu_model(), os.cpu_count())}
            if a.cpu_threads > 1:
                fps_all, n_all, used, wall = cpu_baseline_all_cores(cfg, frames_sets[0][:min(B, 8 * a.cpu_threads)],
                                                                    a.cpu_seconds / 2, bool(M), orbfe.MP_DTYPE, a.cpu_threads)
                out["cpu_baseline"]["all_cores"] = {"value": fps_all, "unit": "frames/s", "cores": used,
                                                    "sample": "%d frames over %d host threads, %.1f s wall" % (n_all, used, wall)}
        emit_line(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    if not all_ok:
        if rank == 0:
            sys.stderr.write("bench.py: the timed path's results differ from the oracle / the gathered bytes from the packed ones: %s %s\n" % (
                json.dumps(verified), json.dumps(gather_verified)))
        sys.exit(4)


if __name__ == "__main__":
    main()
