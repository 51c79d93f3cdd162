#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python3 -m pytest tests/test_gpu_conv_gn.py tests/test_gpu_ncsnpp.py tests/test_gpu_ddpm.py -m gpu -q 2>&1 | tail -5
timeout 600 python3 tools/ab_knob.py natinf_set_conv_gn_wide 1 3 2>&1 | tail -7
for sh in "32 512 256 256 0" "32 512 256 256 256"; do timeout 300 python3 tools/bench_conv_gn.py $sh 2>&1 | grep TFLOP; done
