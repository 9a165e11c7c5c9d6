#!/bin/bash
# GPU box: the evidence set of a round -- kernel-trace stats of the default bench line, and the PMC traffic of the
# dominant kernel of the three propagation workloads (queens-64, queens-128, sudoku-25).  -> gpurun_out/prof_<tag>_*/
tag=${1:-round}
set -x
tools/profile_bench.sh ${tag}_q64 cs_propagate_ne_shave "queens-64 propagation-only, 262144 instances per launch, state-only entry (kernel 7)" -- --no-queens128 --steps 20 --warmup 3
tools/profile_bench.sh ${tag}_q128 cs_propagate_ne_shave "queens-128 propagation-only, 131072 instances per launch, state-only entry (kernel 7)" -- --queens 128 --instances 131072 --steps 20 --warmup 3
tools/profile_bench.sh ${tag}_sud25 cs_propagate_ne_lds "sudoku-25x25 propagation-only, 262144 instances per launch, state-only entry (kernel 2, lists in L2)" -- --sudoku 5 --steps 10 --warmup 2
