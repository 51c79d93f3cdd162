python -m pytest tests/test_gpu_mmdit.py -x -q 2>&1 | tail -1
for d in 0 2; do echo "dbg $d: $(NATINF_FLASH_DBG=$d python tools/bench_flash.py)"; done
echo "w4: $(NATINF_FLASH_W4=1 python tools/bench_flash.py)"; echo "v1: $(NATINF_FLASH_V1=1 python tools/bench_flash.py)"
