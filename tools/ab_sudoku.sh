#!/bin/bash
# A/B of builds of libcsolve_hip.so on the SAME box, sudoku-25 propagation bench (kernel 2): tools/ab/libcsolve_hip_base.so
# and every tools/ab/libcsolve_hip_var*.so against the in-tree one, alternating, three rounds.
set -e
mkdir -p gpurun_out
for round in 1 2 3; do
  for lib in tools/ab/libcsolve_hip_base.so tools/ab/libcsolve_hip_var*.so new; do
    [ "$lib" != new ] && [ ! -e "$lib" ] && continue
    path=""; [ "$lib" != new ] && path="$PWD/$lib"
    CSOLVE_HIP_LIB=$path timeout -k 10 120 python bench.py --sudoku 5 --instances 262144 --no-search --no-cpu 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$(basename $lib .so)', round(r['roofline']['kernel_ms']*1000,2),'us sudoku-25 frac',round(r['roofline']['frac'],3), '8d', round(r['roofline'].get('survey_8d_node_frac',0),3))"
  done
done
