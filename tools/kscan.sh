export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
for MODE in 0 2 3; do for K in 64; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kscan_$K -o k -- python3 $R/tools/bench_gemm.py one 16 8192 1536 $K 0 1 0 50 $MODE > /dev/null 2>&1
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$R/gpurun_out/kscan_$K/k_kernel_stats.csv")) if "k_gemm" in r["Name"]]
for r in rows: print("mode", $MODE, "K", $K, r["Calls"], r["AverageNs"], r["MinNs"])
PY
done; done
