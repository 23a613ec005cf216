#!/bin/bash
# copies the summaries of gpurun_out/prof_<tag> (tools/profile_all.sh) and of the newest --no-overlap trace
# (tools/match_trace.sh) into profiles/ under the tag's name.  Usage: tools/publish_profile.sh <tag>
set -e
tag=${1:?tag}
cd "$(dirname "$0")/.."
cp gpurun_out/prof_$tag/kernels.json profiles/${tag}_kernels.json
cp gpurun_out/prof_$tag/kernel_stats.csv profiles/${tag}_kernel_stats.csv
head=$(cat gpurun_out/prof_$tag/git_head)
f=$(ls -t $(find gpurun_out/match_trace -name "*kernel_stats.csv") | head -1)
python3 - "$f" "$head" "profiles/${tag}_no_overlap_kernel_stats.txt" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = []
for r in rows:
    n = r['Name'].replace('void ', '').replace('orbfe::proj::', '').replace('orbfe::', '').replace('(anonymous namespace)::', '')
    n = re.sub(r'\((orbfe|unsigned|int|PipelineDesc|ProjArgs).*', '', n)
    out.append("%-48s calls %4s avg %9.4f ms min %8.4f max %8.4f sd %7.4f" % (n[:48], r['Calls'], float(r['AverageNs']) / 1e6, float(r['MinNs']) / 1e6, float(r['MaxNs']) / 1e6, float(r['StdDev']) / 1e6))
txt = ("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-io --no-latency --no-overlap "
       "(matcher after the extraction on one stream: kernel times without contention), git %s\n" % sys.argv[2]) + "\n".join(out[:9]) + "\n"
open(sys.argv[3], 'w').write(txt)
print(txt)
PY
