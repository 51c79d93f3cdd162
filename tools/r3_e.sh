#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3e; mkdir -p $O
cd $R
python3 -m pytest tests/test_gpu_conv_gn.py tests/test_gpu_ncsnpp.py tests/test_gpu_accuracy.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
python3 tools/ab_build_knob.py natinf_set_fuse_head 0 1 2>&1 | tail -7
for rep in 1 2; do
  for L in libnatinf.so libnatinf_ws40.so; do
    echo "== $L" | tee -a $O/ab.log
    NATINF_LIB=$R/naturaldiffusion_amd/$L python3 tools/bench_conv_gn.py 2>&1 | grep TFLOP | tee -a $O/ab.log
    NATINF_LIB=$R/naturaldiffusion_amd/$L python3 tools/ab_knob.py natinf_set_gemm_pref512 1 2>&1 | grep "ms per" | tail -2 | tee -a $O/ab.log
  done
done
python3 bench.py --no-sd3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['mean_launch_ms'], d['roofline_gemm']['frac'], d['roofline_whole_denoiser']['frac'])"
