#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3g; mkdir -p $O
cd $R
python3 -m pytest tests/test_gpu_ddpm.py tests/test_gpu_ncsnpp.py -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -E "ddpm per|passed|failed|Error|assert" $O/pytest.log | tail -12
bash tools/trace_fwd.sh 0 1 2>&1 | grep -E "head_conv|total GEMM"
python3 tools/ab_build_knob.py natinf_set_fuse_head 0 1 2>&1 | tail -7
python3 bench.py --no-sd3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['mean_launch_ms'], d['roofline_gemm']['frac'], d['roofline_whole_denoiser']['frac'])"
