#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O
cd $R
python3 -m pytest tests/test_gpu_mmdit.py tests/test_gpu_dit.py tests/test_gpu_bench_multirank.py -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -E "rel_rms|passed|failed|Error|assert" $O/pytest.log | tail -12
python3 bench.py --workload sd3 --steps 2 --warmup 1 > $O/sd3.json 2> $O/sd3.err; python3 -c "
import json; d=json.loads(open('$O/sd3.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_gemm']['frac'], d.get('accuracy'))"
python3 bench.py --workload sd3 --fp8 --steps 2 --warmup 1 --no-cpu-baseline > $O/sd3_fp8.json 2> $O/sd3_fp8.err; python3 -c "
import json; d=json.loads(open('$O/sd3_fp8.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_gemm']['frac'])"
