#!/bin/bash
# per-kernel and per-GEMM-shape times of the NCSN++ forward at B=512 (GPU box): tools/trace_fwd.sh [epilogue mode]
export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_fwd; m=${1:-0}; f=${2:-1}; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/fwd_once.py $m 4 $f > $O.log 2>&1
T=$(find $O -name "*kernel_trace.csv" | head -1); S=$(find $O -name "*kernel_stats.csv" | head -1)
python3 $R/tools/analyze_trace.py $T 512
head -12 $S | cut -c1-140
