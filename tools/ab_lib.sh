#!/bin/bash
# same-box A/B of two builds of the library (NATINF_LIB): tools/ab_lib.sh <lib_a.so> <lib_b.so>
A=${1:-naturaldiffusion_amd/libnatinf.so}; B=${2:-naturaldiffusion_amd/libnatinf_b.so}
for rep in 1 2; do
  for L in $A $B; do
    echo "== $L"
    NATINF_LIB=$PWD/$L python tools/bench_flash.py 2>&1 | tail -2
    NATINF_LIB=$PWD/$L python tools/ab_knob.py natinf_set_gemm_pref512 1 2>&1 | grep "ms per" | tail -1
  done
done
