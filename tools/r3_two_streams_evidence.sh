#!/bin/bash
# GPU box: the unprofiled measurements behind the two-stream order (rocprofv3 serialises the two lanes: profiles/README.md) -> gpurun_out/two_streams/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/two_streams; mkdir -p $O; cd $R
python3 tools/ab_two_batches.py 2>&1 | grep -v amdgpu.ids > $O/ab_two_batches.txt
python3 tools/ab_two_batches_knobs.py 2>&1 | grep -v amdgpu.ids > $O/ab_two_batches_knobs.txt
python3 tools/ab_two_batches_masks.py 2>&1 | grep -v amdgpu.ids > $O/ab_two_batches_masks.txt
python3 tools/debug_det3.py 2>&1 | grep -v amdgpu.ids > $O/two_engines_cu_masks.txt
python3 tools/debug_det5.py 2>&1 | grep -v amdgpu.ids > $O/single_launch_beside_an_engine.txt
python3 tools/soak_two_streams.py 12 2>&1 | grep -v amdgpu.ids > $O/soak_two_streams.txt
for s in 1 2 3; do python3 bench.py --streams $s --steps 12 --warmup 2 --no-single-stream --no-sd3 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py --streams', d['config']['streams'], '--steps 12:', d['value'], 'images/s,', d['ms_per_step'], 'ms per step')"; done > $O/bench_streams.txt
tail -n +1 $O/*.txt | tail -60
