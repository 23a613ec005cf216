cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pyrprof; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-latency --no-match > $out/a.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --output-format csv -d $out/b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-latency --no-match > $out/b.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-latency --no-match > $out/c.log 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
for sub in "abc":
    fs = glob.glob("gpurun_out/pyrprof/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not fs: print(sub, "no csv"); continue
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        acc[r["Kernel_Name"].split("(")[0][-28:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "pyramid" in k or "orient" in k:
            print(sub, k, {c: round(sum(x)/len(x), 1) for c, x in v.items()})
PY
