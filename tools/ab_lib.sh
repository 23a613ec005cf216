# A/B of two builds of the library on ONE box: bench.py --lib <so>, alternating, three rounds
cd "$GRAFT_REPO_ROOT"
for r in 1 2 3; do
  for lib in "$@"; do
    python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-io --no-latency $BENCH_ARGS --lib $lib 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d['roofline']['stage_ms_per_step']
print('$lib: value %.0f ms/step %.3f | pyr %.3f fast %.3f quad %.3f orient %.3f total %.3f match %.3f' % (d['value'], d['ms_per_step'], st['pyramid_resize'], st['fast_nms_blur'], st['quadtree'], st['orient_brief'], st['total'], st.get('match_projection', 0.0)))"
  done
done
