#!/bin/bash
# A/B of builds of libcsolve_hip.so on the SAME box: tools/ab/libcsolve_hip_base.so and every tools/ab/libcsolve_hip_var*.so
# against the in-tree one, alternating, three rounds of the headline bench each (state_only leg: kernel_ms of
# queens-64 and queens-128).  Rank builds only by runs on one device (cdna_hip_programming.md rule 24).
set -e
mkdir -p gpurun_out
for round in 1 2 3; do
  for lib in tools/ab/libcsolve_hip_base.so tools/ab/libcsolve_hip_var*.so new; do
    [ "$lib" != new ] && [ ! -e "$lib" ] && continue
    path=""; [ "$lib" != new ] && path="$PWD/$lib"
    CSOLVE_HIP_LIB=$path timeout -k 10 120 python bench.py --layout intervals --no-sudoku25 --no-search --no-cpu 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$(basename $lib .so)', round(r['roofline']['kernel_ms']*1000,2),'us q64 frac',round(r['roofline']['frac'],3))"
    CSOLVE_HIP_LIB=$path timeout -k 10 120 python bench.py --layout intervals --queens 128 --instances 131072 --no-search --no-cpu 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$(basename $lib .so)', round(r['roofline']['kernel_ms']*1000,2),'us q128 frac',round(r['roofline']['frac'],3))"
  done
done
