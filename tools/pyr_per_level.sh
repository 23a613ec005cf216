#!/bin/bash
# per-dispatch durations of pyramid_kernel (one dispatch per level) from a rocprofv3 kernel trace, with and without the matcher beside
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for tag in nomatch match; do
  out=gpurun_out/pyr_$tag; rm -rf $out; mkdir -p $out
  extra=""; [ $tag = nomatch ] && extra="--no-match"
  rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-io --no-latency $extra > $out/log.txt 2>&1
  python3 - "$out" "$tag" <<'PY'
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 4 steps: group consecutive pyramid dispatches
pyr = [r for r in rows if "pyramid_kernel" in r["Kernel_Name"]]
L = 7
steps = [pyr[i:i + L] for i in range(0, len(pyr), L)][-4:]
for lvl in range(L):
    d = [(int(s[lvl]["End_Timestamp"]) - int(s[lvl]["Start_Timestamp"])) / 1e3 for s in steps]
    gap = [(int(s[lvl + 1]["Start_Timestamp"]) - int(s[lvl]["End_Timestamp"])) / 1e3 for s in steps] if lvl + 1 < L else [0]
    print(sys.argv[2], "level", lvl + 1, "dur us %.1f" % (sum(d) / len(d)), "gap to next us %.1f" % (sum(gap) / len(gap)))
tot = [(int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"])) / 1e3 for s in steps]
print(sys.argv[2], "chain us %.1f" % (sum(tot) / len(tot)))
PY
  find $out -name "*.csv" -size +1M -delete; find $out -name "*.db" -delete
done
