#!/bin/bash
# GPU box: the evidence behind a bench line -- rocprofv3 kernel-trace stats of the default bench command and the two
# PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) of the dominant kernel.
# usage: profile_bench.sh <tag> <kernel substring> <workload text> -- <bench args...>      -> gpurun_out/prof_<tag>/
tag=$1; kern=$2; workload=$3; shift 4
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o run -- python3 $root/bench.py --no-cpu --no-search "$@" > $out/bench_under_trace.json 2> $out/err.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o run -- python3 $root/bench.py --no-cpu --no-search --no-graph "$@" > /dev/null 2>> $out/err.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o run -- python3 $root/bench.py --no-cpu --no-search --no-graph "$@" > /dev/null 2>> $out/err.txt
cd $root
python3 tools/pmc_traffic.py $out/fetch $out/write "$kern" "$workload" "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py --no-cpu --no-graph $*" $out/pmc_traffic.json
cp $out/trace/run_kernel_stats.csv $out/kernel_stats.csv 2>/dev/null || find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
head -5 $out/kernel_stats.csv
