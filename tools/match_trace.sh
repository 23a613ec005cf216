cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/match_trace; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-io --no-latency --no-overlap > $out/t.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1); cut -d, -f1-8 $f | sed 's/(orbfe::proj::ProjArgs)//; s/void orbfe:://; s/(anonymous namespace):://' | cut -c1-150 | head -12
