#!/bin/bash
# per-stage instruction counts of fast_blur_kernel via the timing-only ablation build (make ablation; ORBFE_FAST_MODE
# is read by liborbfe_ablation.so only)
# usage: fast_stage_counters.sh [modes...]   (default: 0 1 6 10 2 3)
out=gpurun_out/pmc_fast; rm -rf $out; mkdir -p $out
modes="${@:-0 1 6 10 2 3}"
make -s -C orb_slam3_v1.0_amd/csrc ablation
ABL=orb_slam3_v1.0_amd/csrc/liborbfe_ablation.so
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in $modes; do
  ORBFE_FAST_MODE=$m rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $out/m$m -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-io --no-latency --no-match --lib $ABL > $out/m$m.log 2>&1
done
