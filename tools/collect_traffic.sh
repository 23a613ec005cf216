#!/bin/bash
# Runs on the GPU box (via gpurun): PMC passes for HBM traffic, one counter per pass, no tracing domains.
# Usage: tools/collect_traffic.sh <outdir>
set -e
out=${1:-gpurun_out/pmc}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/calib_$c" -- tools/pmc_calib.bin > "$out/calib_$c.log" 2>&1
  rocprofv3 --pmc $c --output-format csv -d "$out/bench_$c" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-latency > "$out/bench_$c.log" 2>&1
done
find "$out" -name "*counter_collection.csv" | head
