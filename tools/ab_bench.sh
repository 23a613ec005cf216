#!/bin/bash
# A/B on ONE box: bench.py with every library under tools/ab/ (earlier builds) and with the shipped one, alternating, the
# headline steps plus a 3 s sustained region each.  Usage: tools/ab_bench.sh <out-prefix> <rounds> [bench args...]
out=${1:?prefix}; rounds=${2:-3}; shift; shift || true
args="${@:---steps 20 --warmup 5 --no-cpu-baseline --no-host-io --no-latency --sustained-seconds 3}"
for i in $(seq 1 $rounds); do
  for lib in tools/ab/*.so shipped; do
    name=$(basename $lib .so); name=${name#liborbfe_}
    if [ "$lib" = shipped ]; then python3 bench.py $args > ${out}_${name}_$i.json 2>/dev/null || exit 1
    else python3 bench.py $args --lib $lib > ${out}_${name}_$i.json 2>/dev/null || exit 1; fi
    python3 - "${out}_${name}_$i.json" "$name" "$i" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
s = d.get("sustained") or {}
st = d["roofline"]["stage_ms_per_step"]
print("%-8s round %s: headline %.0f  sustained %.0f  fast %.4f ms  quadtree %.4f ms  clock %.0f MHz  verified %s" % (
    sys.argv[2], sys.argv[3], d["value"], s.get("value", 0), st["fast_nms_blur"], st["quadtree"],
    (s.get("sclk_mhz") or {}).get("mean") or 0, (d.get("verified") or {}).get("kp_desc_equal")), flush=True)
PY
  done
done
