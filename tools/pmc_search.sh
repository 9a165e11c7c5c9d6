#!/bin/bash
# GPU box: two rocprofv3 --pmc passes of SQ counters over ONE search (bench.py --workload search); sums every counter over
# the launches of the kernels whose name contains <kernel substring> and prints it per launched child / per node.
# usage: pmc_search.sh <out name> <kernel substring> -- <bench args...>     (counters only: no other trace domains)
name=$1; kern=$2; shift 3
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$name
rm -rf $out ${out}b; mkdir -p $out ${out}b
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $out -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload search --steps 1 --warmup 0 "$@" > $out/bench.json 2> $out/err.txt
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d ${out}b -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload search --steps 1 --warmup 0 "$@" > ${out}b/bench.json 2>> $out/err.txt
python3 - $out $kern <<'PY'
import csv, glob, json, sys, collections
out, kern = sys.argv[1], sys.argv[2]
cfg = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])["config"]
nodes, cuts = cfg["nodes"], cfg["cuts"]
print(f"nodes {nodes} cuts {cuts} solutions {cfg['solutions']} iterations {cfg['iterations']}")
for d in (out, out + "b"):
    tot = collections.defaultdict(float); cnt = collections.defaultdict(int); dur = 0.0; seen = set()
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if kern in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
                if r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"]); dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print(f"-- {kern}: {len(seen)} launches, {dur:.2f} ms under the counters")
    for k in sorted(tot):
        print(f"{k:24s} total {tot[k]:18.0f}  per node {tot[k] / nodes:9.3f}")
PY
