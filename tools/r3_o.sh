#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3o; mkdir -p $O
cd $R
timeout 600 python3 tools/bench_conv_gn.py 2>&1 | grep TFLOP > $O/isolated.txt
for sh in "4 512 256 256 0" "4 512 512 256 0" "4 512 256 256 256"; do timeout 300 python3 tools/bench_conv_gn.py $sh 2>&1 | grep TFLOP >> $O/isolated.txt; done
cat $O/isolated.txt
bash tools/trace_fwd.sh 0 1 > $O/by_shape.txt 2>&1
head -50 $O/by_shape.txt
