"""Calibration only (never on the product path): rocBLAS / hipBLASLt bf16 GEMM rates through torch.matmul on the engine's plain
GEMM shapes, beside this library's own kernel (GPU box)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.argv = sys.argv[:1]
import tools.bench_gemm as BG   # noqa: E402  (SHAPES is emptied on import)

def lib_rate(M, N, K, iters=10):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    for _ in range(3): c = a @ b.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): c = a @ b.t()
    e1.record(); torch.cuda.synchronize()
    return 2.0 * M * N * K / (e0.elapsed_time(e1) / iters) / 1e9

shapes = [(32768, 1536, 1536), (32768, 6144, 1536), (32768, 1536, 6144), (65536, 1152, 1152), (65536, 4608, 1152), (8192, 8192, 8192), (4096, 4096, 4096),
          (131072, 256, 2304), (524288, 128, 1152), (131072, 512, 256)]
print(f"{'M,N,K':>24} {'torch (BLAS lib)':>18} {'this library':>14}")
for (M, N, K) in shapes:
    mine = BG.run(0, M, N, K, 0, 1, 0, iters=10)[1]
    print(f"{str((M, N, K)):>24} {lib_rate(M, N, K):13.0f} TF/s {mine:9.0f} TF/s", flush=True)
