#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3u; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_ncsnpp.py tests/test_gpu_ddpm.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -25 $O/pytest.log
timeout 600 python3 tools/ab_build_knob.py natinf_set_attn_qkv 0 1 2>&1 | tail -7
