cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_ob; rm -rf $out; mkdir -p $out
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-latency --no-match > $out/c.log 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
f = glob.glob("gpurun_out/pmc_ob/c/**/*counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-40:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k] += 1
for k, v in acc.items():
    if "orbfe" in k or "kernel" in k:
        d = n[k] / max(len(v), 1)
        print(k, {a: round(b / d / 1e6, 2) for a, b in v.items()}, "M per launch; launches", int(d))
PY
