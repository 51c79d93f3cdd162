for v in 3 4; do echo "variant $v: $(NATINF_FLASH_VARIANT=$v python -m pytest tests/test_gpu_mmdit.py -x -q -k 'flash' 2>&1 | tail -1)"; done
for v in 0 3 4 0 3 4; do echo "variant $v: $(NATINF_FLASH_VARIANT=$v python tools/bench_flash.py)"; done
