#!/bin/bash
# K-scan of one 256x256 tile per CU (M=65536, N=256): launch time vs K for the full epilogue (0), no epilogue (2 -> c_mode 102)
# and everything-but-stores (3 -> c_mode 103).  usage: tools/kscan.sh [variant]
V=${1:-16}
for cm in 0 2 3; do
  for K in 64 256 1024 4096; do
    python tools/bench_gemm.py one $V 65536 256 $K 0 1 0 50 $cm
  done
done
