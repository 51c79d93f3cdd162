#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3h; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_conv_gn.py -m gpu -q -x > $O/pytest_cg.log 2>&1; echo "pytest rc=$?" >> $O/pytest_cg.log; tail -12 $O/pytest_cg.log
timeout 900 python3 -m pytest tests/test_gpu_ncsnpp.py tests/test_gpu_ddpm.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log
timeout 300 python3 tools/bench_conv_gn.py 8 512 256 256 0; timeout 300 python3 tools/bench_conv_gn.py 8 512 512 256 0; timeout 300 python3 tools/bench_conv_gn.py 8 512 256 256 256
timeout 600 python3 tools/ab_build_knob.py natinf_set_fuse_gn8 0 1 2>&1 | tail -7
