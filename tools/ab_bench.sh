#!/bin/bash
# A/B of two builds of libcsolve_hip.so on the SAME box: tools/ab/libcsolve_hip_base.so against the in-tree one,
# alternating, three rounds of the headline bench each (state_only leg: kernel_ms of queens-64 and queens-128).
set -e
mkdir -p gpurun_out
for round in 1 2 3; do
  for which in base new; do
    lib=""; [ $which = base ] && lib="$PWD/tools/ab/libcsolve_hip_base.so"
    CSOLVE_HIP_LIB=$lib timeout -k 10 120 python bench.py --layout intervals --no-sudoku25 --no-search 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which', round(r['roofline']['kernel_ms']*1000,2),'us q64 frac',round(r['roofline']['frac'],3),'| q128', round(r['queens128']['roofline']['kernel_ms']*1000,2),'us frac',round(r['queens128']['roofline']['frac'],3))"
  done
done
