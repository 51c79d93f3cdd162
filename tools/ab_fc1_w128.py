"""fc1 of the MMDiT fp8 path, (32768, 6144, 1536) with the tanh-GELU + e4m3 + E8M0 epilogue, on the eight-wave tile (natinf_set_gemm_w128 0 / 2) and on the four-wave tile (1), same process."""
import sys, ctypes as C
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from naturaldiffusion_amd._lib import lib, check, stream_ptr, ptr
dev = torch.device("cuda:0"); torch.manual_seed(0)
def quant(x):
    q = torch.empty(x.shape, dtype=torch.uint8, device=dev); s = torch.empty(x.shape[0], dtype=torch.float32, device=dev)
    check(lib.natinf_debug_quant_fp8_rows(ptr(x), ptr(q), ptr(s), x.shape[0], x.shape[1], stream_ptr()), "quant"); return q, s
def run_mx(M, N, K, iters=20):
    qa = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev)
    b = torch.randn(N, K, device=dev) * 0.05; qb, sb = quant(b)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    c8 = torch.empty(M, N, dtype=torch.uint8, device=dev); cm = torch.empty(M * N // 32 + 1024, dtype=torch.uint8, device=dev)
    res = {}
    for name, args in (("plain->bf16", lambda it: (M, N, K, ptr(qa), None, None, ptr(qb), ptr(sb), None, ptr(c), None, 0, it, stream_ptr())),
                       ("plain->fp8mx", lambda it: (M, N, K, ptr(qa), None, None, ptr(qb), ptr(sb), None, ptr(c8), ptr(cm), 3, it, stream_ptr())),
                       ("plain->gelu+fp8mx (fc1)", lambda it: (M, N, K, ptr(qa), None, None, ptr(qb), ptr(sb), None, ptr(c8), ptr(cm), 3 | (2 << 8), it, stream_ptr()))):
        check(lib.natinf_debug_gemm_fp8(*args(2)), "gemm_fp8"); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); check(lib.natinf_debug_gemm_fp8(*args(iters)), "gemm_fp8"); e1.record(); torch.cuda.synchronize()
        res[name] = round(2.0 * M * N * K / (e0.elapsed_time(e1) / iters) / 1e9)
    return res
for w in (0, 1, 2, 0, 1, 2):
    check(lib.natinf_set_gemm_w128(w), "set")
    print(f"w128={w}", run_mx(32768, 6144, 1536), flush=True)
