#!/bin/bash
# Runs on the GPU box (gpurun): one validate_k4.py process per ISA variant of the failing kernel
# (tools/k4_fault_isa_variants.py built them), optionally the bit-level diagnosis of one variant and the
# stand-alone probe.  usage: k4_fault_run_variants.sh OUTFILE [variant ...]   (no variant: all)
set -u
out=$PWD/gpurun_out/${1:-k4_variants.txt}
shift
mkdir -p gpurun_out
: > $out
if [ -x tools/lds_addr_war ]; then
  echo "=== stand-alone probe tools/lds_addr_war" | tee -a $out
  timeout -k 10 120 tools/lds_addr_war 2>&1 | tee -a $out || exit 1
fi
cd tools/fault_wt
libs=""
if [ $# -eq 0 ]; then libs=$(ls variants/libcsolve_hip_*.so); else for v in "$@"; do libs="$libs variants/libcsolve_hip_$v.so"; done; fi
for lib in $libs; do
  name=${lib#variants/libcsolve_hip_}; name=${name%.so}
  echo "=== $name" | tee -a $out
  CSOLVE_HIP_LIB=$PWD/$lib timeout -k 10 240 python tools/validate_k4.py 64 262144 10 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $out || exit 1
done
