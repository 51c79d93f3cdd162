#!/bin/bash
# run on the GPU box:  bash tools/profile_sd3.sh <tag> [--fp8]  -- rocprofv3 kernel stats of one SD3 28-step batch + the bench line
TAG=${1:-g}; EXTRA=${2:-}
export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profile_$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o sd3 -- python3 $R/bench.py --workload sd3 $EXTRA --steps 1 --warmup 0 --no-roofline --no-cpu-baseline > $O.stats.log 2>&1
python3 $R/bench.py --workload sd3 $EXTRA --steps 2 --warmup 1 > $O.bench.json 2> $O.bench.err
tail -1 $O.bench.json | cut -c1-1500
