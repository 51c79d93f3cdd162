#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python3 -m pytest tests/test_gpu_conv_gn.py tests/test_gpu_ncsnpp.py tests/test_gpu_ddpm.py -m gpu -q 2>&1 | tail -3
timeout 600 python3 tools/ab_knob.py natinf_set_conv_gn_tpb 1 2 4 2>&1 | tail -10
