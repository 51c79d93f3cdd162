#!/bin/bash
# run on the GPU box:  bash tools/profile_sd3.sh <tag> [--fp8]
# 1) rocprofv3 --kernel-trace --stats of one SD3 28-step batch, 2) two separate --pmc passes (FETCH_SIZE, WRITE_SIZE: TCC slots do not fit both) over two
# forwards of 8 sequences (tools/sd3_fwd_once.py) for HBM traffic per kernel, 3) the bench line.  tools/summarize_profile.py <tag> rNN copies the summaries.
TAG=${1:-g}; EXTRA=${2:-}
FP8=""; [ "$EXTRA" = "--fp8" ] && FP8="fp8"
export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profile_$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o sd3 -- python3 $R/bench.py --workload sd3 $EXTRA --steps 1 --warmup 0 --no-roofline --no-cpu-baseline > $O.stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/sd3_fwd_once.py $FP8 > $O.fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/sd3_fwd_once.py $FP8 > $O.write.log 2>&1
python3 $R/bench.py --workload sd3 $EXTRA --steps 2 --warmup 1 > $O.bench.json 2> $O.bench.err
tail -1 $O.bench.json | cut -c1-1500
