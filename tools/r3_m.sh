#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3m; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_conv_gn.py tests/test_gpu_ncsnpp.py tests/test_gpu_ddpm.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -25 $O/pytest.log
for sh in "4 512 256 256 0" "4 512 512 256 0" "4 512 256 256 256" "4 512 512 256 512"; do timeout 300 python3 tools/bench_conv_gn.py $sh 2>&1 | grep TFLOP; done
timeout 600 python3 tools/ab_build_knob.py natinf_set_fuse_gn4 0 1 2>&1 | tail -8
