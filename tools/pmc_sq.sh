#!/bin/bash
# GPU box: one rocprofv3 --pmc pass of SQ counters over a bench command; prints per-node averages of the timed kernel.
# usage: pmc_sq.sh <out name> <kernel substring> <nodes per launch> -- <bench args...>
name=$1; kern=$2; nodes=$3; shift 4
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$name
rm -rf $out ${out}b; mkdir -p $out ${out}b
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $out -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-search --no-graph "$@" > $out/bench.json 2> $out/err.txt
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d ${out}b -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-search --no-graph "$@" > ${out}b/bench.json 2>> $out/err.txt
python3 - $out $kern $nodes <<'PY'
import csv, glob, sys, collections
out, kern, nodes = sys.argv[1], sys.argv[2], float(sys.argv[3])
for d in (out, out + "b"):
    per = collections.defaultdict(list)
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if kern in r["Kernel_Name"]:
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(per):
        v = sorted(per[k]); med = v[len(v) // 2]
        print(f"{k:24s} launches {len(v):4d} median {med:16.0f} per node {med / nodes:10.2f}")
PY
