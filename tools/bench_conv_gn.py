"""Isolated timing of k_conv_gn2 / k_conv_gn (fused GroupNorm-apply + SiLU + 3x3 conv; env REGW=0: the LDS-ring kernel): bench_conv_gn.py [res B cin N c1 [iters]] ...
With no arguments: the engine's shapes at B = 512.  Prints ms and TFLOP/s (2*M*N*(9*cin + c1))."""
import os, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd._lib import lib, check, ptr, stream_ptr

def run(res, B, cin, N, c1, iters=20):
    dev = "cuda"
    M = B * res * res
    x = torch.randn(B, res, res, cin, device=dev).bfloat16()
    sc = torch.rand(B, cin, device=dev) + 0.5; sh = torch.randn(B, cin, device=dev) * 0.3
    w = (torch.randn(N, 9 * cin + c1, device=dev) / (9 * cin) ** 0.5).bfloat16()
    a1 = torch.randn(M, c1, device=dev).bfloat16() if c1 else None
    bias = torch.randn(N, device=dev); out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    part = torch.zeros(M // min(64, res * res), N // 4, 2, device=dev)          # (one row per tile; per sample at res 8 / 4)
    wf = torch.zeros_like(w) if os.environ.get('REGW', '1') != '0' else None
    args = (res, B, N, cin, c1, ptr(x), ptr(sc), ptr(sh), ptr(w), ptr(wf), ptr(a1), ptr(bias), None, 0.7071, ptr(out), ptr(part))
    check(lib.natinf_debug_conv_gn(*args, 3, stream_ptr()), "warm")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    check(lib.natinf_debug_conv_gn(*args, iters, stream_ptr()), "run")
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    fl = 2.0 * M * N * (9 * cin + c1)
    print(f"res {res} B {B} cin {cin} N {N} c1 {c1}: {dt * 1e3:.3f} ms  {fl / dt / 1e12:.1f} TFLOP/s", flush=True)

if len(sys.argv) > 1:
    a = [int(v) for v in sys.argv[1:]]
    run(*a)
else:
    for spec in ((32, 512, 128, 128, 0), (32, 512, 256, 128, 0), (32, 512, 128, 128, 256), (32, 512, 384, 128, 0), (16, 512, 256, 256, 0),
                 (16, 512, 512, 256, 0), (16, 512, 256, 256, 512), (16, 512, 128, 256, 0), (8, 512, 256, 256, 0), (8, 512, 512, 256, 0), (8, 512, 256, 256, 256)):
        run(*spec)
