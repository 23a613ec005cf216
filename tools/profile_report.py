#!/usr/bin/env python3
"""Summarise the passes of tools/profile_all.sh into ONE per-kernel JSON (+ the rocprofv3 stats CSV).

Per kernel of namespace orbfe:
  avg_ms / min / max / stddev / calls       rocprofv3 --kernel-trace --stats
  fetch/write/hbm bytes per launch          --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes); KiB -> bytes, and the
                                            gfx950 factor measured on tools/pmc_calib.bin for 4 B/lane streams
                                            (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reads half)
  valu/salu/lds/vmem/smem per wave, waves   --pmc SQ_INSTS_* / SQ_WAVES
  issue_frac                                waves x valu_per_wave / (1024 SIMDs x 2.4 GHz / 2 cycles x avg duration)
  active/wait shares                        SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES
  valu_busy                                 SQ_ACTIVE_INST_VALU (quad-cycles a wave spends on a VALU instruction, summed) x 4 /
                                            (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): the share of the SIMDs' time that goes into
                                            vector instructions -- rocprofv3's VALUBusy (counter_defs.yaml) per kernel
  effective_clock_mhz                       GRBM_GUI_ACTIVE / 8 / the kernel's average duration (MI355X_MICROARCH.md, DVFS
                                            give-back; reads high on dispatches shorter than ~0.3 ms)
  valu_cycles_per_instruction_weighted      tools/isa_mix.py: the kernel's emitted ISA priced with profiles/r03_valu_rate.txt
Usage: profile_report.py <dir of profile_all.sh> <tag>"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python"))
from orbfe.provenance import source_sha  # noqa: E402

root, tag = sys.argv[1], sys.argv[2]
SIMDS, CLK_HZ = 1024, 2.4e9  # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, max clock; one wave64 VALU op = 2 cycles


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


def newest(sub, pattern):
    fs = glob.glob(os.path.join(root, sub, "**", pattern), recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None


def counters(sub):
    f = newest(sub, "*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


out = {"_meta": {"tag": tag, "source_sha": source_sha(),
                 "git_head": open(os.path.join(root, "git_head")).read().strip() if os.path.exists(os.path.join(root, "git_head")) else None,
                 "command": open(os.path.join(root, "command")).read().strip() if os.path.exists(os.path.join(root, "command")) else None,
                 "issue_peak": "1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction = 1.2288e12 wave-instructions/s"}}
for name in ("bench_plain.log", "trace.log"):
    try:
        for line in open(os.path.join(root, name)):
            if line.startswith("{"):
                d = json.loads(line)
                out["_meta"][name.split(".")[0]] = {"value": d["value"], "ms_per_step": d["ms_per_step"],
                                                    "frames_per_step": d["config"]["frames_per_step"],
                                                    "workload": d["config"]["workload"].split(" ")[0],
                                                    "stage_ms_per_step": d["roofline"]["stage_ms_per_step"]}
    except OSError:
        pass

kern = defaultdict(dict)
f = newest("trace", "*kernel_stats.csv")
if f:
    shutil.copy(f, os.path.join(root, "kernel_stats.csv"))
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if "orbfe" not in k:
            continue
        kern[k].update(calls=int(r["Calls"]), avg_ms=float(r["AverageNs"]) / 1e6, min_ms=float(r["MinNs"]) / 1e6,
                       max_ms=float(r["MaxNs"]) / 1e6, stddev_ms=float(r["StdDev"]) / 1e6, pct_gpu_time=float(r["Percentage"]))

GiB = float(1 << 30)
calib = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    k = counters("calib_" + c)
    for pat in ("copy4", "copy16"):
        v = [x for kn, cs in k.items() if pat in kn for x in cs.get(c, [])]
        if v:
            calib["%s/%s" % (c, pat)] = GiB / (sum(v) / len(v) * 1024.0)
out["_meta"]["pmc_calibration_true_bytes_per_counter_KiB_x1024"] = calib
fetch, write = counters("bench_FETCH_SIZE"), counters("bench_WRITE_SIZE")
for k in set(fetch) | set(write):
    if "orbfe" not in k:
        continue
    fv, wv = fetch.get(k, {}).get("FETCH_SIZE", []), write.get(k, {}).get("WRITE_SIZE", [])
    fb = sum(fv) / max(1, len(fv)) * 1024.0 * calib.get("FETCH_SIZE/copy4", 2.0)
    wb = sum(wv) / max(1, len(wv)) * 1024.0 * calib.get("WRITE_SIZE/copy4", 1.0)
    kern[k].update(fetch_bytes_per_launch=fb, write_bytes_per_launch=wb, hbm_bytes_per_launch=fb + wb)

sqa, sqb = counters("sq_a"), counters("sq_b")
for k, cs in sqa.items():
    if "orbfe" not in k or not cs.get("SQ_WAVES"):
        continue
    n = len(cs["SQ_WAVES"])
    waves = sum(cs["SQ_WAVES"]) / n
    per = {c: sum(v) / n / waves for c, v in cs.items() if c.startswith("SQ_INSTS_")}
    kern[k].update(waves_per_launch=waves, valu_per_wave=per.get("SQ_INSTS_VALU"), salu_per_wave=per.get("SQ_INSTS_SALU"),
                   lds_per_wave=per.get("SQ_INSTS_LDS"), vmem_rd_per_wave=per.get("SQ_INSTS_VMEM_RD"),
                   vmem_wr_per_wave=per.get("SQ_INSTS_VMEM_WR"), smem_per_wave=per.get("SQ_INSTS_SMEM"))
    if "avg_ms" in kern[k] and per.get("SQ_INSTS_VALU"):
        kern[k]["issue_frac"] = waves * per["SQ_INSTS_VALU"] / (SIMDS * CLK_HZ / 2.0 * kern[k]["avg_ms"] * 1e-3)
for k, cs in sqb.items():
    if "orbfe" not in k or not cs.get("SQ_WAVE_CYCLES"):
        continue
    wc = sum(cs["SQ_WAVE_CYCLES"])
    for c, key in (("SQ_ACTIVE_INST_ANY", "active_inst_share"), ("SQ_ACTIVE_INST_VALU", "active_valu_share"),
                   ("SQ_ACTIVE_INST_SCA", "active_scalar_share"), ("SQ_WAIT_INST_ANY", "wait_inst_share"),
                   ("SQ_WAIT_ANY", "wait_any_share")):
        if cs.get(c):
            kern[k][key] = sum(cs[c]) / wc
grbm = counters("grbm")
for k, cs in grbm.items():
    if "orbfe" not in k or not cs.get("GRBM_GUI_ACTIVE") or not cs.get("SQ_ACTIVE_INST_VALU"):
        continue
    n = len(cs["GRBM_GUI_ACTIVE"])
    gui = sum(cs["GRBM_GUI_ACTIVE"]) / n / 8.0  # rocprofv3 sums the 8 XCDs
    kern[k]["gui_cycles_per_launch"] = gui
    kern[k]["valu_busy"] = sum(cs["SQ_ACTIVE_INST_VALU"]) / len(cs["SQ_ACTIVE_INST_VALU"]) * 4.0 / (SIMDS * gui)
    if "avg_ms" in kern[k]:
        kern[k]["effective_clock_mhz"] = gui / (kern[k]["avg_ms"] * 1e-3) / 1e6

# the emitted ISA of the kernels the bench line prices (tools/isa_mix.py; hipcc is on the box)
ISA = {"fast_blur_kernel<3>": ("kernels_fast.hip", "fast_blur_kernelILi3E"), "orient_brief_kernel": ("kernels_desc.hip", "orient_brief_kernel"),
       "pyramid_kernel": ("kernels_pyramid.hip", "14pyramid_kernel"), "quadtree_kernel<512, false, 512>": ("kernels_quadtree.hip", "quadtree_kernelILi512ELb0ELi512E")}
try:
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_mix
    for k in kern:
        for pat, (f, sym) in ISA.items():
            if k.endswith(pat):
                try:
                    mx = isa_mix.mix(os.path.join(ROOT, "orb_slam3_v1.0_amd", "csrc", f), sym)
                except (SystemExit, OSError, subprocess.CalledProcessError) as e:
                    kern[k]["isa_mix_error"] = str(e)[:200]
                    continue
                kern[k]["valu_cycles_per_instruction_weighted"] = mx["valu_cycles_per_instruction_weighted"]
                kern[k]["isa_mix"] = {q: mx[q] for q in ("valu_static", "valu_static_fast_group", "valu_static_slow_group",
                                                        "valu_static_priced_by_assumption", "salu_static", "lds_static")}
                if kern[k].get("valu_per_wave") and kern[k].get("waves_per_launch") and kern[k].get("gui_cycles_per_launch"):
                    # the dynamic count priced at the static mean, over the SIMD cycles the launch had
                    kern[k]["issue_frac_weighted"] = (kern[k]["waves_per_launch"] * kern[k]["valu_per_wave"] * mx["valu_cycles_per_instruction_weighted"] /
                                                      (SIMDS * kern[k]["gui_cycles_per_launch"]))
except ImportError:
    pass
out["kernels"] = {k: kern[k] for k in sorted(kern, key=lambda k: -kern[k].get("pct_gpu_time", 0))}
json.dump(out, open(os.path.join(root, "kernels.json"), "w"), indent=1)
for k, v in out["kernels"].items():
    print("%-44s avg %.3f ms  hbm %7.1f MB  valu/wave %6.0f salu %5.0f lds %4.0f  issue %.2f  weighted %.2f  valu_busy %.2f  clk %4.0f MHz" % (
        k[-44:], v.get("avg_ms", 0), v.get("hbm_bytes_per_launch", 0) / 1e6, v.get("valu_per_wave") or 0,
        v.get("salu_per_wave") or 0, v.get("lds_per_wave") or 0, v.get("issue_frac") or 0, v.get("issue_frac_weighted") or 0,
        v.get("valu_busy") or 0, v.get("effective_clock_mhz") or 0))
