#!/bin/bash
# GPU box: the evidence set of a round -- kernel-trace stats of the bench commands, the PMC traffic of the dominant kernel
# of the three propagation workloads (queens-64, queens-128, sudoku-25), kernel-trace stats and SQ counters of the fused
# ALL search, and the drop-in's call latency.  -> gpurun_out/prof_<tag>_*/, gpurun_out/<tag>_*
tag=${1:-round}
set -x
tools/profile_bench.sh ${tag}_q64 cs_propagate_ne_shave "queens-64 propagation-only, 2097152 instances per launch, state-only entry (kernel 7)" -- --no-queens128 --no-sudoku25 --steps 20 --warmup 3
tools/profile_bench.sh ${tag}_q128 cs_propagate_ne_shave "queens-128 propagation-only, 524288 instances per launch, state-only entry (kernel 7)" -- --queens 128 --instances 524288 --steps 20 --warmup 3
tools/profile_bench.sh ${tag}_sud25 cs_propagate_ne_lds "sudoku-25x25 propagation-only, 524288 instances per launch, state-only entry (kernel 2, 40 % givens)" -- --sudoku 5 --instances 524288 --steps 10 --warmup 2
tools/trace_fused.sh 16 ${tag}_search
tools/trace_fused.sh 17 ${tag}_search
tools/pmc_search.sh ${tag}_q16 cs_step_packed -- --search-queens 16 > gpurun_out/${tag}_search/pmc_step_packed_q16.txt 2>&1
python3 tools/time_dropin_call.py 64 > gpurun_out/${tag}_dropin_call_latency.txt 2>&1
python3 tools/time_dropin_call.py 128 >> gpurun_out/${tag}_dropin_call_latency.txt 2>&1
