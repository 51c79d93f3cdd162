#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3l; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_conv_gn.py tests/test_gpu_ncsnpp.py tests/test_gpu_ddpm.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
for sh in "8 512 256 256 0" "8 512 512 256 0" "8 512 256 256 256"; do timeout 300 python3 tools/bench_conv_gn.py $sh 2>&1 | grep TFLOP; done
timeout 600 python3 tools/ab_knob.py natinf_set_conv_gn8_tile 0 1 2>&1 | tail -8
