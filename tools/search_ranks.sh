#!/bin/bash
# Rehearsal of the sharded search on ONE GPU: 1 rank, then 4 ranks sharing cuda:0 over gloo (host tensors between
# the ranks), queens-N ALL and a schedule MIN model.  Writes the bench lines (with per-rank clocks) to
# gpurun_out/search_ranks.jsonl.   usage: tools/search_ranks.sh [queens N] [schedule T]
set -e
Q=${1:-15}; T=${2:-9}
out=gpurun_out/search_ranks.jsonl; mkdir -p gpurun_out; : > $out
run1() { timeout -k 10 300 python bench.py --workload search --gpus 1 "$@" | grep '^{' >> $out; }
run4() { timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 \
          --master-port 29533 bench.py --workload search --gpus 4 --comm gloo --same-device "$@" 2>/dev/null | grep '^{' >> $out; }
run1 --search-queens $Q
run4 --search-queens $Q
run4 --search-queens $Q --slice 64
run1 --search-schedule $T
run4 --search-schedule $T
run4 --search-schedule $T --slice 256
echo done
