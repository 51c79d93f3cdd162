#!/bin/bash
# same-box A/B of two builds of the library (NATINF_LIB): tools/ab_lib.sh <lib_a.so> <lib_b.so> -- <command ...>
# (build the second one with  make -C naturaldiffusion_amd/csrc BUILD=$PWD/naturaldiffusion_amd/csrc/build_b OUT=$PWD/naturaldiffusion_amd/libnatinf_b.so [EXTRA=...])
A=$1; B=$2; shift 3
for rep in 1 2; do
  for L in $A $B; do
    echo "== $L"
    NATINF_LIB=$PWD/$L "$@" 2>&1 | tail -2
  done
done
