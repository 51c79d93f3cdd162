#!/bin/bash
# round-3 first GPU pass: GPU suite, diffusers probe, the default bench line (all three workloads), flash-attention event/rocprof reconciliation
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3a; mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
python3 tools/capture_diffusers.py > $O/capture.log 2>&1; cat $O/capture.log
python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/sd3trace -o sd3 -- python3 $R/bench.py --workload sd3 --steps 1 --warmup 0 --no-cpu-baseline > $O/sd3trace.json 2> $O/sd3trace.err
python3 - <<PY
import csv, glob
f = glob.glob("$O/sd3trace/**/*kernel_trace.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "k_flash_attn64" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("flash launches", len(d), "in-engine mean us", sum(d[:-6]) / max(1, len(d) - 6), "isolated (last 6) us", d[-6:])
PY
