#!/bin/bash
# GPU box: kernel-trace stats of the search workload.  usage: trace_search.sh <tag> -- <bench args...>  -> gpurun_out/trace_<tag>/
tag=$1; shift 2
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/trace_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -o run -- python3 $root/bench.py --workload search "$@" > $out/bench.json 2> $out/err.txt
find $out/t -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
cd $root
python3 - $out/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f'{float(r["TotalDurationNs"]) / tot * 100:5.1f}%  {int(r["Calls"]):7d} calls  avg {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Name"][:90]}')
PY
tail -c 400 $out/bench.json
