#!/bin/bash
# LDS-pipe counters of fast_blur_kernel per ablation mode (is the LDS array a co-bottleneck of the VALU?)
out=gpurun_out/pmc_fast_lds; rm -rf $out; mkdir -p $out
modes="${@:-0 6 10 3}"
ABL=${ABL:-orb_slam3_v1.0_amd/csrc/liborbfe_ablation.so}
[ -f "$ABL" ] || make -s -C orb_slam3_v1.0_amd/csrc ablation
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in $modes; do
  ORBFE_FAST_MODE=$m rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/m$m -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-io --no-latency --no-match --lib $ABL > $out/m$m.log 2>&1
done
python3 - <<'PY'
import csv, glob
from collections import defaultdict
for m in (0, 1, 6, 10, 2, 3):
    fs = glob.glob("gpurun_out/pmc_fast_lds/m%d/**/*counter_collection.csv" % m, recursive=True)
    if not fs: continue
    acc = defaultdict(float); n = 0
    for r in csv.DictReader(open(fs[0])):
        if "fast_blur_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n += 1
    w = acc["SQ_WAVES"] or 1
    print("mode %2d" % m, {k: round(v / w, 2) for k, v in acc.items() if k != "SQ_WAVES"}, "waves", int(w), "dispatches", n // 8)
PY
