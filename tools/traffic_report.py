#!/usr/bin/env python3
"""Turn the PMC passes of tools/collect_traffic.sh into per-launch HBM bytes per kernel.

FETCH_SIZE / WRITE_SIZE are in KiB (cdna_hip_programming.md section 7).  On gfx950 FETCH_SIZE is only
calibrated for 16 B/lane streams (it reads exactly half); for the 4 B/lane pattern of the ORB kernels the
factor is measured here from tools/pmc_calib.bin (1 GiB read + 1 GiB written per kernel)."""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"


def per_kernel(sub, counter):
    f = max(glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            acc[name.split("(")[0]].append(float(r["Counter_Value"]))
    return acc


GiB = float(1 << 30)
calib = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    k = per_kernel("calib_" + c, c)
    for name in ("copy4", "copy16"):
        v = [x for kn, xs in k.items() if name in kn for x in xs]
        calib[(c, name)] = GiB / (sum(v) / len(v) * 1024.0)  # true bytes per reported byte
print("calibration (true bytes / (counter x 1024)):", {"%s/%s" % k: round(v, 3) for k, v in calib.items()})
out = {"calibration": {"%s/%s" % k: v for k, v in calib.items()}, "kernels": {}}
fetch = per_kernel("bench_FETCH_SIZE", "FETCH_SIZE")
write = per_kernel("bench_WRITE_SIZE", "WRITE_SIZE")
for kn in sorted(set(fetch) | set(write)):
    if "orbfe" not in kn:
        continue
    fb = sum(fetch.get(kn, [0])) / max(1, len(fetch.get(kn, [0]))) * 1024.0 * calib[("FETCH_SIZE", "copy4")]
    wb = sum(write.get(kn, [0])) / max(1, len(write.get(kn, [0]))) * 1024.0 * calib[("WRITE_SIZE", "copy4")]
    out["kernels"][kn] = {"launches": len(fetch.get(kn, [])), "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                          "hbm_bytes_per_launch": fb + wb}
    print("%-70s n=%3d fetch=%8.1f MB write=%8.1f MB" % (kn[-70:], len(fetch.get(kn, [])), fb / 1e6, wb / 1e6))
frames_per_launch, workload = 512, "euroc_752x480"  # bench.py defaults; overridden by the bench line of the profiled run
try:
    for line in open(os.path.join(root, "bench_FETCH_SIZE.log")):
        if line.startswith("{"):
            cfg = json.loads(line)["config"]
            frames_per_launch = int(cfg["frames_per_step"])
            workload = cfg["workload"].split(" ")[0]
except (OSError, ValueError, KeyError):
    pass
out["_meta"] = {"workload": workload, "frames_per_launch": frames_per_launch,
                "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline",
                "correction": "FETCH_SIZE x1024 x%.1f, WRITE_SIZE x1024 x%.1f (measured with tools/pmc_calib.bin, 4 B/lane and 16 B/lane streams)"
                              % (calib[("FETCH_SIZE", "copy4")], calib[("WRITE_SIZE", "copy4")])}
json.dump(out, open(os.path.join(root, "traffic.json"), "w"), indent=1)
