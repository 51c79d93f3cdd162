#!/bin/bash
# N=128 convolution shapes: 512x128 hand-pipelined tile (13) vs the 4-wave 256x128 ring (9)
for v in 9 13; do
  python tools/bench_gemm.py one $v 524288 128 1152 0 9 32 20
  python tools/bench_gemm.py one $v 524288 128 2304 0 9 32 20
  python tools/bench_gemm.py one $v 524288 128 1152 256 9 32 20
done
