#!/bin/bash
export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3k; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/fwd_once.py 0 4 1 > $O.log 2>&1
S=$(find $O -name "*kernel_stats.csv" | head -1)
grep -E "attn|k_gn_|splitk|head_conv|stem" $S | cut -c1-150
head -8 $S | cut -c1-150
