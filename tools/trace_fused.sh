#!/bin/bash
# per-iteration trace of the fused ALL search (CSGPU_SEARCH_TRACE) and a kernel trace of the same search
#   usage: tools/trace_fused.sh <queens N> <out dir under gpurun_out>
set -e
N=${1:-16}; out=gpurun_out/${2:-fused}; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
CSGPU_SEARCH_TRACE=1 timeout -k 10 120 python3 $root/bench.py --workload search --search-queens $N --steps 1 --warmup 0 > $root/$out/trace_q$N.json 2> $root/$out/trace_q$N.err
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/kt -o run -- python3 $root/bench.py --workload search --search-queens $N --steps 2 --warmup 1 > $root/$out/kt_q$N.json 2> $root/$out/kt_q$N.err
cd $root
f=$(ls $out/kt/run_kernel_stats.csv $out/kt/*/run_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_q$N.csv
rm -rf $out/kt
echo done
