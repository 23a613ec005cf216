#!/bin/bash
# SQ counters of the projection-matcher kernels (bench workload, matcher not overlapped)
out=gpurun_out/pmc_match; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $out/a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-io --no-latency --no-overlap > $out/a.log 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
f = glob.glob("gpurun_out/pmc_match/a/**/*counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    w = v.get("SQ_WAVES", 1) or 1
    print(k, {c: round(x / w, 1) for c, x in v.items() if c != "SQ_WAVES"}, "waves", int(w))
PY
