#!/bin/bash
# per-stage instruction counts of fast_blur_kernel via the timing-only ablation modes (ORBFE_FAST_MODE)
out=gpurun_out/pmc_fast; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in 0 1 6 10 2 3; do
  ORBFE_FAST_MODE=$m rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $out/m$m -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-match > $out/m$m.log 2>&1
done
