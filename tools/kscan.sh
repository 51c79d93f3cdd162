python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c75-110
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c75-110
python bench.py --workload sd3 --steps 2 --warmup 1 2>&1 | tail -1 | cut -c75-110
