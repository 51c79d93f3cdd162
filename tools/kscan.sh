python -m pytest tests/test_gpu_ncsnpp.py tests/test_gpu_dit.py -x -q 2>&1 | tail -2
for i in 1 2; do
echo "fused:   $(python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c75-100)"
echo "unfused: $(NATINF_NCSNPP_UNFUSED_ATTN=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c75-100)"
done
