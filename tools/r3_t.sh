#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python3 -m pytest tests/test_gpu_conv_gn.py tests/test_gpu_ncsnpp.py tests/test_gpu_ddpm.py -m gpu -q 2>&1 | tail -3
bash tools/ab_lib3.sh gpurun_in/libnatinf_old.so gpurun_in/libnatinf_new.so 2>&1 | grep "=="
for L in old new; do echo "-- $L"; for sh in "32 512 128 128 0" "32 512 256 128 0" "16 512 256 256 0" "8 512 256 256 0"; do NATINF_LIB=$R/gpurun_in/libnatinf_$L.so timeout 300 python3 tools/bench_conv_gn.py $sh 2>&1 | grep TFLOP; done; done
