"""torch.matmul (rocBLAS / hipBLASLt) bf16 GEMM in a loop for ~10 s: the vendor library's sustained rate, for tools/power_probe.sh."""
import sys, time, torch
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8192, 8192, 8192)
a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
for _ in range(5): c = a @ b.t()
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 10.0:
    for _ in range(50): c = a @ b.t()
    torch.cuda.synchronize(); n += 50
dt = time.perf_counter() - t0
print(f"torch.matmul ({M}, {N}, {K}): {2.0 * M * N * K * n / dt / 1e12:.0f} TFLOP/s sustained over {dt:.1f} s")
