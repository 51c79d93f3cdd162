#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
bash tools/trace_fwd.sh 0 1 > $O/by_shape.txt 2>&1
head -60 $O/by_shape.txt | tail -16
S=$(find $R/gpurun_out/trace_fwd -name "*kernel_stats.csv" | head -1); head -30 $S | cut -c1-150 > $O/stats_head.txt; cat $O/stats_head.txt | tail -22
