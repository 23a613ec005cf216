#!/bin/bash
# Runs on the GPU box (via gpurun): every profile the bench line's roofline is derived from, on ONE command.
#   1. rocprofv3 --kernel-trace --stats            -> per-kernel average duration, over >= 200 steps ($TRACE_STEPS) of the timed loop
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE      -> HBM traffic (separate passes, calibrated with tools/pmc_calib.bin)
#   3. rocprofv3 --pmc SQ_* (two passes of <= 8)    -> instructions per wave, issue / wait shares
#   4. rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -> VALU-busy share of the SIMDs and the effective clock
# The program follows `--` directly (no env / bash -c hop).  No tracing domain is combined with --pmc.
# Usage: tools/profile_all.sh <tag> <git-head> [bench args...]
set -e
tag=${1:?tag}; head=${2:-unknown}; shift; shift || true
args="${@:---steps 5 --warmup 2 --no-cpu-baseline --no-host-io --no-latency --sustained-seconds 0 --no-verify}"
# the kernel-trace pass runs the timed loop for TRACE_STEPS steps (the counter passes serialise the kernels: a few steps there)
trace_args=$(echo "$args" | sed -E "s/--steps [0-9]+/--steps ${TRACE_STEPS:-200}/")
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
[ -x tools/pmc_calib.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/pmc_calib.bin tools/pmc_calib.hip
echo "$head" > "$out/git_head"
echo "python3 bench.py $args" > "$out/command"
python3 bench.py $args > "$out/bench_plain.log" 2>&1
echo "plain bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py $trace_args > "$out/trace.log" 2>&1
echo "kernel trace done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/calib_$c" -- tools/pmc_calib.bin > "$out/calib_$c.log" 2>&1
  rocprofv3 --pmc $c --output-format csv -d "$out/bench_$c" -- python3 bench.py $args > "$out/bench_$c.log" 2>&1
  echo "$c done"
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES \
  --output-format csv -d "$out/sq_a" -- python3 bench.py $args > "$out/sq_a.log" 2>&1
echo "SQ pass a done"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA \
  --output-format csv -d "$out/sq_b" -- python3 bench.py $args > "$out/sq_b.log" 2>&1
echo "SQ pass b done"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES \
  --output-format csv -d "$out/grbm" -- python3 bench.py $args > "$out/grbm.log" 2>&1
echo "GRBM pass done"
# keep the merged-back directory small: the per-dispatch CSVs are summarised here, on the box
python3 tools/profile_report.py "$out" "$tag"
find "$out" -name "*.csv" -size +2M -delete
find "$out" -name "*.db" -delete
ls "$out"
