#!/bin/bash
# same-box per-kernel comparison of the two GEMM epilogues on the NCSN++ forward (GPU box): tools/ab_trace.sh
export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_trace
for m in 1 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/m$m -- python3 $R/tools/fwd_once.py $m 4 > $O.m$m.log 2>&1
  T=$(find $O/m$m -name "*kernel_trace.csv" | head -1)
  S=$(find $O/m$m -name "*kernel_stats.csv" | head -1)
  echo "== epilogue mode $m"; python3 $R/tools/analyze_trace.py $T 512 | head -24
  head -12 $S | cut -c1-150
done
