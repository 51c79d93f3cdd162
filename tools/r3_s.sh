#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3s; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_ncsnpp.py tests/test_gpu_ddpm.py -m gpu -q 2>&1 | tail -3
rm -rf $R/gpurun_out/trace_fwd
bash tools/trace_fwd.sh 0 1 > $O/by_shape.txt 2>&1
S=$(find $R/gpurun_out/trace_fwd -name "*kernel_stats.csv" | head -1); head -14 $S | cut -c1-150
timeout 300 python3 tools/ab_knob.py natinf_set_attn256 1 1 2>&1 | tail -3
