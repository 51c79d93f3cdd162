#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; mkdir -p $O
cd $R
python3 -m pytest tests/test_gpu_inception.py tests/test_gpu_conv_gn.py tests/test_gpu_ncsnpp.py tests/test_gpu_gemm_epilogue.py -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -E "inception|passed|failed|Error|assert" $O/pytest.log | tail -30
python3 bench.py --workload sd3 --steps 1 --warmup 1 --no-cpu-baseline > $O/sd3.json 2> $O/sd3.err; python3 -c "
import json; d=json.loads(open('$O/sd3.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline'])"
