python -m pytest tests/test_gpu_mmdit.py -x -q 2>&1 | tail -3
python tools/bench_mmdit.py 8 2>&1 | grep '"ms"'
python tools/bench_mmdit.py 8 fp8 2>&1 | grep '"ms"'
python bench.py --workload sd3 --steps 2 --warmup 1 2>&1 | tail -1 | cut -c75-110
python bench.py --workload sd3 --fp8 --steps 2 --warmup 1 2>&1 | tail -1 > gpurun_out/sd3_fp8.json; cut -c75-110 gpurun_out/sd3_fp8.json
