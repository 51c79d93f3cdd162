"""k_gemm_w128 (variant 29: one wave per SIMD, 128x128 wave tiles) beside the two-waves-per-SIMD 256x256 tile (variant 26) and the vendor library
(torch.matmul: calibration only) on the transformer engines' plain GEMM shapes (GPU box).  usage: bench_w128.py [iters]"""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sys.argv = sys.argv[:1]
import tools.bench_gemm as BG   # noqa: E402


def lib_rate(M, N, K):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    for _ in range(3): c = a @ b.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): c = a @ b.t()
    e1.record(); torch.cuda.synchronize()
    return 2.0 * M * N * K / (e0.elapsed_time(e1) / iters) / 1e9


shapes = [(32768, 1536, 1536), (32768, 6144, 1536), (32768, 1536, 6144), (32768, 4608, 1536), (65536, 1152, 1152), (65536, 4608, 1152), (65536, 1152, 4608),
          (8192, 8192, 8192), (4096, 4096, 4096), (131072, 512, 256), (131072, 256, 256)]
print(f"{'M,N,K':>24} {'vendor':>8} {'v26':>8} {'v29':>8} {'v29/vendor':>10}   (TFLOP/s)   err")
for (M, N, K) in shapes:
    r26 = BG.run(26, M, N, K, 0, 1, 0, iters=iters)[1]
    ms, r29, err = BG.run(29, M, N, K, 0, 1, 0, iters=iters, check_ref=(M * N <= 1 << 28))
    rv = lib_rate(M, N, K)
    print(f"{str((M, N, K)):>24} {rv:8.0f} {r26:8.0f} {r29:8.0f} {r29 / rv:10.3f}   {ms * 1e3:7.1f} us   {err}", flush=True)
