#!/bin/bash
# run on the GPU box:  bash tools/profile_round.sh <tag>
# 1) rocprofv3 --kernel-trace --stats of the default bench command (1 step), 2) two separate --pmc passes
# (FETCH_SIZE, WRITE_SIZE: TCC slots do not fit both) for HBM traffic per kernel.
TAG=${1:-c}
export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profile_$TAG
mkdir -p $O
CMD="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-sd3 --no-fid50k --no-validate"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > $O.stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMD > $O.fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMD > $O.write.log 2>&1
python3 $R/bench.py > $O.bench.json 2> $O.bench.err
tail -1 $O.bench.json | cut -c1-400
