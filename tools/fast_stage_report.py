#!/usr/bin/env python3
"""Per-wave instruction counts of fast_blur_kernel per ablation mode (tools/fast_stage_counters.sh)."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_fast"
for m in (0, 1, 6, 10, 2, 3):
    fs = glob.glob(os.path.join(root, "m%d" % m, "**", "*counter_collection.csv"), recursive=True)
    fs.sort(key=os.path.getmtime, reverse=True)
    if not fs:
        continue
    acc = defaultdict(float); n = 0
    for r in csv.DictReader(open(fs[0])):
        if "fast_blur_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    w = acc["SQ_WAVES"] or 1
    print("mode %2d" % m, {k: round(v / w, 1) for k, v in acc.items() if k != "SQ_WAVES"}, "waves", int(w))
