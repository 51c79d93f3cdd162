#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3n; mkdir -p $O
cd $R
timeout 600 python3 -m pytest tests/test_gpu_conv_gn.py -m gpu -q 2>&1 | tail -3
timeout 600 python3 tools/ab_knob.py natinf_set_conv_gn_warm 0 1 3 7 15 2>&1 | tail -16
timeout 600 python3 tools/ab_build_knob.py natinf_set_fuse_gn4 0 1 2>&1 | tail -7
