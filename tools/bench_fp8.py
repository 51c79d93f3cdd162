"""fp8 GEMM (k_gemm_fp8): correctness against the dequantised fp32 product and speed against the bf16 kernel (GPU box)."""
import sys, ctypes as C
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd._lib import lib, check, stream_ptr, ptr
dev = torch.device("cuda:0")
torch.manual_seed(0)

def quant(x):
    q = torch.empty(x.shape, dtype=torch.uint8, device=dev); s = torch.empty(x.shape[0], dtype=torch.float32, device=dev)
    check(lib.natinf_debug_quant_fp8_rows(ptr(x), ptr(q), ptr(s), x.shape[0], x.shape[1], stream_ptr()), "quant")
    return q, s

def run(M, N, K, iters=10, chk=False):
    a = torch.randn(M, K, device=dev) * (torch.rand(M, 1, device=dev) * 3 + 0.1); b = torch.randn(N, K, device=dev) * 0.05
    bias = torch.randn(N, device=dev)
    qa, sa = quant(a); qb, sb = quant(b)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    args = lambda it: (M, N, K, ptr(qa), ptr(sa), None, ptr(qb), ptr(sb), ptr(bias), ptr(c), None, 0, it, stream_ptr())
    check(lib.natinf_debug_gemm_fp8(*args(2)), "gemm_fp8"); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); check(lib.natinf_debug_gemm_fp8(*args(iters)), "gemm_fp8"); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    out = {"tf": 2.0 * M * N * K / ms / 1e9, "us": ms * 1e3}
    if chk:
        da = qa.view(torch.float8_e4m3fn).float() * sa[:, None]; db = qb.view(torch.float8_e4m3fn).float() * sb[:, None]
        ref_q = da @ db.t() + bias                      # what the kernel must reproduce (same quantised operands)
        ref = a @ b.t() + bias                          # what fp8 costs against the unquantised product
        out["err_vs_dequant"] = ((c.float() - ref_q).abs().max() / ref_q.abs().max()).item()
        out["err_vs_fp32"] = ((c.float() - ref).abs().max() / ref.abs().max()).item()
        tq = da[:4].cpu(); out["quant_ok"] = bool(torch.allclose(tq, a[:4].cpu(), rtol=0.07, atol=1e-3))
    return out
print(run(512, 256, 256, 2, True)); print(run(1000, 520, 384, 2, True)); print(run(4096, 1536, 1536, 2, True))
for shp in ((32768, 1536, 1536), (32768, 3072, 1536), (32768, 6144, 1536), (32768, 1536, 6144), (8192, 8192, 8192)):
    print(shp, {k: round(v, 1) for k, v in run(*shp).items()})

def run_mx(M, N, K, iters=10):
    qa = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev); ma = torch.randint(120, 130, (M * K // 32 + 1024,), dtype=torch.uint8, device=dev)
    b = torch.randn(N, K, device=dev) * 0.05; qb, sb = quant(b)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    c8 = torch.empty(M, N, dtype=torch.uint8, device=dev); cm = torch.empty(M * N // 32 + 1024, dtype=torch.uint8, device=dev)
    res = {}
    for name, args in (("mxa->bf16", lambda it: (M, N, K, ptr(qa), None, ptr(ma), ptr(qb), ptr(sb), None, ptr(c), None, 0, it, stream_ptr())),
                       ("plain->bf16", lambda it: (M, N, K, ptr(qa), None, None, ptr(qb), ptr(sb), None, ptr(c), None, 0, it, stream_ptr())),
                       ("plain->fp8mx", lambda it: (M, N, K, ptr(qa), None, None, ptr(qb), ptr(sb), None, ptr(c8), ptr(cm), 3, it, stream_ptr()))):
        check(lib.natinf_debug_gemm_fp8(*args(2)), "gemm_fp8"); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); check(lib.natinf_debug_gemm_fp8(*args(iters)), "gemm_fp8"); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        res[name] = round(2.0 * M * N * K / ms / 1e9)
    return res
for w in (0, 1, 0, 1):          # the eight-wave 256x256 tile (k_gemm_fp8) against the four-wave one (k_gemm_w128_fp8), same process
    check(lib.natinf_set_gemm_w128(w), "set")
    for shp in ((32768, 1536, 1536), (32768, 1536, 6144), (32768, 6144, 1536), (32768, 4608, 1536)):
        print(f"w128={w} mx", shp, run_mx(*shp), flush=True)
