#!/bin/bash
# Rehearsal of the sharded search on ONE GPU: 1 rank, then 4 ranks sharing cuda:0 over gloo (host tensors between
# the ranks; bench.py starts its own ranks), queens-N ALL, queens-128 ANY (BASELINE configs[3] instance) and a schedule
# MIN model (configs[4] shape).  Writes the bench lines (with per-rank clocks) to gpurun_out/search_ranks.jsonl.
#   usage: tools/search_ranks.sh [queens N] [schedule T]
set -e
Q=${1:-15}; T=${2:-12}
out=gpurun_out/search_ranks.jsonl; mkdir -p gpurun_out; : > $out
run1() { timeout -k 10 300 python bench.py --workload search --gpus 1 "$@" | grep '^{' >> $out; }
run4() { timeout -k 10 300 python bench.py --workload search --gpus 4 --comm gloo --same-device "$@" 2>/dev/null | grep '^{' >> $out; }
run1 --search-queens $Q
run4 --search-queens $Q
run1 --search-queens 128 --search-objective ANY
run4 --search-queens 128 --search-objective ANY
run1 --search-schedule $T --steps 1 --warmup 0
run4 --search-schedule $T --steps 1 --warmup 0
echo done
