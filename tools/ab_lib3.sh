#!/bin/bash
# same-box comparison of several builds of the library on the NCSN++ forward: tools/ab_lib3.sh lib1.so lib2.so ...
for rep in 1 2; do
  for L in "$@"; do
    echo "== $L: $(NATINF_LIB=$PWD/$L python tools/ab_knob.py natinf_set_gemm_pref512 1 2>&1 | grep 'ms per' | tail -1)"
  done
done
