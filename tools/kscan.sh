python -m pytest tests/test_gpu_ncsnpp.py -x -q 2>&1 | tail -3
for b in 8 32 128 512; do echo "batch $b plain: $(python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c75-135)"; echo "batch $b graph: $(python bench.py --batch $b --graph --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c75-135)"; done
