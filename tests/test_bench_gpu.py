"""bench.py end to end on the GPU at toy sizes: the JSON line the driver parses must carry every contracted field
(metric / value / n_gpus / roofline / cpu_baseline / latency / config.gather_bytes_per_step ...), in every mode the docs
name: default, --images DIR, C4 strong scaling with --emulate-world, --force-gather (a 1-rank RCCL group)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(args, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_default_line_has_the_contracted_fields(built):
    d = run_bench(["--steps", "3", "--warmup", "1", "--batch", "48", "--cpu-seconds", "1", "--cpu-threads", "0", "--latency-calls", "20",
                   "--frame-sets", "2", "--sustained-seconds", "0.5", "--latency-tail-calls", "200"])
    assert d["unit"] == "frames/s" and d["value"] > 1000 and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "u8" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None and "752x480" in d["metric"] and abs(d["ms_per_step"] - 48 * 1e3 / d["value"]) < 1e-6 * d["ms_per_step"] + 1e-9
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["kernel"] == "fast_nms_blur"
    assert set(r["stage_ms_per_step"]) >= {"pyramid_resize", "fast_nms_blur", "quadtree", "orient_brief", "total", "match_projection"}
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 1 and c["unit"] == "frames/s"
    lat = d["latency"]
    for k in ("extract_pageable_ms", "extract_pinned_ms", "extract_oracle_ms", "match_projection_ms", "match_projection_oracle_ms",
              "prepare_and_extract_pinned_ms", "prepare_and_extract_oracle_ms"):
        assert lat[k] > 0, k
    assert lat["extract_pinned_ms"] < lat["extract_oracle_ms"]
    # round 4: the per-frame chain as three calls and as one submission, checked against the oracle inside the bench itself
    for k in ("project_map_points_ms", "three_calls_ms", "track_frame_ms", "track_frame_pageable_ms", "track_frame_oracle_ms"):
        assert lat[k] > 0, k
    assert lat["track_frame_equals_oracle"] is True and lat["track_frame_matches"] > 100
    assert lat["track_frame_map_ms"] > 0 and lat["track_frame_map_equals_track_frame"] is True
    assert lat["track_frame_ms"] < lat["track_frame_oracle_ms"]
    assert lat["track_initialization_ms"] > 0 and lat["track_initialization_equals_oracle"] is True and lat["track_initialization_matches"] > 50
    cfg = d["config"]
    assert cfg["frames_per_step"] == 48 and cfg["gather"] == "none" and cfg["gather_bytes_per_step"] == 0 and "workload" in cfg
    assert d["value_host_io"] > 0 and d["value_host_io_pageable"] > 0 and d["value_host_io_match"] > 0
    assert d["host_io"]["mean_matches_per_frame_host_io_match"] > 100  # the ring really matched (map points by id, resident map)
    assert d["ms_per_step_ranks"]["min"] == d["ms_per_step_ranks"]["max"] == d["ms_per_step"]
    # round 5: the line verifies the path it timed (8 frames of the last step against the oracle), carries a second, longer timed
    # region with the per-step distribution and the clock measured in the kernel, and the tail latency of the per-frame chains
    v = d["verified"]
    assert v["frames"] == 8 and v["kp_desc_equal"] is True and v["match_equal"] is True and len(v["frame_indices"]) == 8
    su = d["sustained"]
    assert su["steps"] >= 200 and su["seconds"] >= 0.5 and su["value"] > 1000
    ms = su["ms_per_step"]
    assert 0 < ms["min"] <= ms["p50"] <= ms["p99"] <= ms["max"] and su["first_100_steps_value"] > 0 and su["last_100_steps_value"] > 0
    assert 0 < ms["window8"]["p50"] <= ms["window8"]["p99"] <= ms["window8"]["max"] <= ms["max"] and ms["max_after_first_step"] <= ms["max"]
    clk = su["sclk_mhz"]
    assert clk["probes"] >= 32 and 100 < clk["min"] <= clk["mean"] <= clk["max"] < 2600, clk  # MI355X: 2400 MHz max clock
    assert 0.5 < su["headline_over_sustained"] < 2.0
    tail = lat["tail"]
    for mode in ("alone", "loaded", "loaded_high_priority"):
        for k in ("track_frame", "track_reference_keyframe"):
            t = tail[mode][k]
            assert t["calls"] == 200 and 0 < t["p50"] <= t["p99"] <= t["max"], (mode, k, t)
    assert tail["mapping_rounds_per_s"] > 1 and lat["track_reference_keyframe_matches"] > 100
    assert lat["track_frame_ms_p99"]["alone"] == tail["alone"]["track_frame"]["p99"]


def test_s0_workload_caps_fire_and_results_stay_exact(built):
    """--workload euroc_752x480_s0: SURVEY S0's nFastFeatures = 16 x nFeatures, the per-level candidate cap of
    src/ORBextractor.cc:449-482 inside the timed path; the line verifies its own batch against the oracle."""
    d = run_bench(["--workload", "euroc_752x480_s0", "--steps", "2", "--warmup", "1", "--batch", "32", "--frame-sets", "1",
                   "--no-cpu-baseline", "--no-host-io", "--no-latency", "--sustained-seconds", "0"])
    assert "nFast=16000" in d["config"]["workload"] and d["verified"]["kp_desc_equal"] and d["verified"]["match_equal"]
    assert "sustained" not in d


def test_images_directory_mode(built, tmp_path):
    from PIL import Image
    from orbfe import synth
    for i, f in enumerate(synth.stream(752, 480, 6, index0=77)):
        Image.fromarray(f).save(tmp_path / ("%04d.pgm" % i))
    d = run_bench(["--steps", "2", "--warmup", "1", "--batch", "16", "--frame-sets", "1", "--images", str(tmp_path), "--no-cpu-baseline",
                   "--no-host-io", "--no-latency", "--sustained-seconds", "0.3"])
    assert d["verified"]["kp_desc_equal"] and d["verified"]["match_equal"]  # image files, too, are checked against the oracle
    assert d["data"].startswith("images: 6 files of") and d["value"] > 1000
    assert d["config"]["mean_keypoints_per_frame"] > 900  # the files went through the extractor (the stream fills its budget)


def test_strong_scaling_emulation_and_one_rank_gather(built):
    e = run_bench(["--workload", "batched_1280x720", "--emulate-world", "8", "--steps", "2", "--warmup", "1", "--frame-sets", "1",
                   "--no-cpu-baseline", "--no-host-io", "--no-latency", "--sustained-seconds", "0.3"])
    assert e["verified"]["kp_desc_equal"] and e["verified"]["match_equal"]
    assert e["scaling"] == "strong" and e["n_gpus"] == 1 and e["config"]["frames_per_step"] == 64 and e["config"]["gather"] == "none"
    assert e["emulate_world"]["world"] == 8 and abs(e["emulate_world"]["predicted_value_at_world"] - 8 * e["value"]) < 1e-6 * e["value"] * 8
    g = run_bench(["--steps", "2", "--warmup", "1", "--batch", "32", "--frame-sets", "1", "--force-gather", "--no-cpu-baseline",
                   "--no-host-io", "--no-latency", "--sustained-seconds", "0.3"])
    assert g["n_gpus"] == 1 and g["config"]["gather"].startswith("rccl all_gather of the per-frame") and g["config"]["gather_bytes_per_step"] == 32 * 8
    assert g["gather_verified"] == {"mismatching_slots_all_ranks": 0, "identical_on_all_ranks": True, "ok": True}
    assert g["verified"]["ranks_failing"] == 0 and len(g["sustained"]["per_rank_p50_ms"]) == 1
    f = run_bench(["--steps", "2", "--warmup", "1", "--batch", "32", "--frame-sets", "1", "--force-gather", "--gather", "full",
                   "--no-cpu-baseline", "--no-host-io", "--no-latency", "--sustained-seconds", "0.3"])
    cap = 1000 + 3 * 8
    assert f["config"]["gather_bytes_per_step"] == 32 * (cap * 60 + 4)
    # the gathered slots of this rank hold exactly the packed keypoints + descriptors + match indices of the last step
    assert f["gather_verified"]["ok"] is True and f["verified"]["kp_desc_equal"] and f["verified"]["match_equal"]
