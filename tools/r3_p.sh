#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
export NATINF_LIB=$R/gpurun_in/libnatinf_dev.so
for sh in "4 512 256 256 0" "4 512 512 256 0" "8 512 256 256 0" "16 512 256 256 0" "32 512 128 128 0"; do echo "== $sh"; timeout 300 python3 tools/conv_gn_timeline.py $sh 2>&1 | grep -v amdgpu.ids; done
