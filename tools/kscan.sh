export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gnprof -o k -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$R/gpurun_out/gnprof/k_kernel_stats.csv")))[:8]: print(r["Name"][:50], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
