"""Where one 256x256 tile of k_gemm_w128<7> (direct fp32 residual-stream epilogue: out = resid + gate * (acc + bias), in place) spends its time, next to the packed bf16
epilogue on the same shape: shader-clock stamps of block 0 (a -DNATINF_DEV library: NATINF_LIB=naturaldiffusion_amd/libnatinf_dev.so) (GPU box)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd._lib import lib, check, ptr, stream_ptr
ts = torch.zeros(16, dtype=torch.int64, device="cuda")
for (M, N, K) in [(32768, 1536, 1536), (32768, 1536, 6144)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16() * 0.02
    bias = torch.zeros(N, device="cuda"); gate = torch.ones(M // 4096, N, device="cuda")
    for mode in ("e7", "e1"):
        x = torch.randn(M, N, device="cuda"); c16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        def run():
            if mode == "e7":
                check(lib.natinf_debug_gemm_fused(0, M, N, K, ptr(a), ptr(b), ptr(bias), None, None, ptr(gate), 12, None, ptr(x), 1.0, 0, ptr(x), 1, None, None, 0, stream_ptr()), "e7")
            else:
                check(lib.natinf_debug_gemm_fused(0, M, N, K, ptr(a), ptr(b), ptr(bias), None, None, None, 30, None, None, 1.0, 0, ptr(c16), 0, None, None, 0, stream_ptr()), "e1")
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        lib.natinf_debug_timestamps(ts.data_ptr()); ts.zero_(); run(); torch.cuda.synchronize(); lib.natinf_debug_timestamps(None)
        t = ts.cpu().tolist()
        print(f"{(M, N, K)} {mode}: {us:.1f} us/launch = {2.0 * M * N * K / us / 1e6:.0f} TF/s; block 0 (100 MHz ticks? raw s_memtime): start->tile0 {t[1]-t[0]}, K loop {t[5]-t[1]}, "
              f"epilogue first half {t[6]-t[5]}, second half (+ store drain for e7) {t[7]-t[6]}; total {t[7]-t[0]}", flush=True)
