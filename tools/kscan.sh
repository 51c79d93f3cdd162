python bench.py --workload sd3 --steps 2 --warmup 1 2>&1 | tail -1 | cut -c1-160
python bench.py --workload sd3 --fp8 --steps 2 --warmup 1 2>&1 | tail -1 > gpurun_out/sd3_fp8.json; cut -c1-160 gpurun_out/sd3_fp8.json
python bench.py --workload sd3 --steps 2 --warmup 1 2>&1 | tail -1 | cut -c1-160
