#!/bin/bash
# per-stage time of fast_blur_kernel via the timing-only ablation build (make ablation; liborbfe_ablation.so reads ORBFE_FAST_MODE):
# 0 = staging only, 1 = +blur, 6 = staging + FAST stage A, 10 = + stage B, 2 = FAST without blur, 3 = product
ABL=${ABL:-orb_slam3_v1.0_amd/csrc/liborbfe_ablation.so}   # ABL=<other timing-only build> for an A/B of two kernels
[ -f "$ABL" ] || make -s -C orb_slam3_v1.0_amd/csrc ablation
for m in ${MODES:-0 1 6 10 2 3}; do
  ORBFE_FAST_MODE=$m python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-io --no-latency --no-match --lib $ABL 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('mode $m', d['roofline']['stage_ms_per_step'])"
done
