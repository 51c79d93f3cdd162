#!/bin/bash
# full GPU suite + the default bench line + smoke
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3i; mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 bench.py > $O/bench.json 2> $O/bench.err; python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d.get('single_stream',{}).get('value'), d['ms_per_step'], d['roofline']['frac'], d['roofline']['mean_launch_ms'], d['roofline_gemm']['frac'], d['roofline_whole_denoiser']['frac'], d['sd3']['value'], d['sd3_fp8']['value'], d['cpu_baseline']['value'])"
