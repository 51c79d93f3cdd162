#!/bin/bash
# SQ counters of k_conv_gn* on one shape (GPU box): tools/pmc_conv_gn.sh "32 512 256 128 0" [tag]   (env REGW=0: k_conv_gn instead of k_conv_gn2)
export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT; spec=${1:-"32 512 256 128 0"}; tag=$(echo $spec | tr ' ' '_'); O=$R/gpurun_out/pmc_cg/${tag}${2:+_$2}; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq1 -- python3 $R/tools/bench_conv_gn.py $spec 3 > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $O/sq2 -- python3 $R/tools/bench_conv_gn.py $spec 3 > $O/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INST_LEVEL_LDS --output-format csv -d $O/sq3 -- python3 $R/tools/bench_conv_gn.py $spec 3 > $O/sq3.log 2>&1
python3 - <<PY
import csv, glob
from collections import defaultdict
for d in ("sq1", "sq2", "sq3"):
    fs = glob.glob("$O/" + d + "/*/*counter_collection.csv")
    if not fs: print(d, "no data"); continue
    agg = defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_conv_gn" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
