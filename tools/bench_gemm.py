"""Micro-benchmark of the GEMM kernel variants on the engine's dominant layer shapes (GPU box).
usage: bench_gemm.py [variant ids ...]      (default: all DMA / ring variants)"""
import sys, ctypes as C
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd._lib import lib, check, stream_ptr

NAMES = {0: "auto", 1: "generic128", 2: "dma256x256", 3: "dma256x128", 4: "dma128x128", 5: "ring256x256", 6: "ring256x128", 7: "ring128x128", 8: "ring64x128", 9: "ring256x128w4", 10: "dma256x128w4", 11: "dma256x256s", 12: "dma128x128s", 13: "dma512x128", 14: "patch256x256", 15: "patch256x128", 16: "dma256x256p", 17: "dma128x128p", 18: "dma256x128w4p", 19: "gemm8ph", 20: "gemm8ph_np", 21: "gemm8ph_rf", 22: "gemm8ph_nprf", 23: "fp8", 24: "abl_nodma", 25: "abl_nomfma", 26: "dma256x256h", 27: "dma512x128h", 28: "conv_gn", 29: "w128_256x256", 30: "w128_schA", 31: "w128_prefetch", 32: "w128_no_dma"}
dev = torch.device("cuda:0")
torch.manual_seed(0)

CMODE = 0
def run(variant, M, N, K0, K1, taps, res, iters=20, check_ref=False):
    C0 = K0 // taps
    if taps == 9:
        Bn = M // (res * res)
        a = torch.zeros(Bn, res + 2, res + 2, C0, dtype=torch.bfloat16, device=dev)
        a[:, 1:-1, 1:-1] = torch.randn(Bn, res, res, C0, device=dev).to(torch.bfloat16)
        logW = res.bit_length() - 1
    else:
        a = torch.randn(M, C0, device=dev).to(torch.bfloat16); logW = 0
    a1 = torch.randn(M, K1, device=dev).to(torch.bfloat16) if K1 else None
    b = (torch.randn(N, K0 + K1, device=dev) * 0.05).to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    args = lambda it: (variant, M, N, K0, K1, taps, logW, 1, a.data_ptr(), a1.data_ptr() if K1 else None, b.data_ptr(), None,
                       c.data_ptr(), CMODE, 1.0, it, stream_ptr())
    check(lib.natinf_debug_gemm(*args(3)), "debug_gemm")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); check(lib.natinf_debug_gemm(*args(iters)), "debug_gemm"); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    err = None
    if check_ref and taps == 1:
        ref = a.float() @ b[:, :K0].float().t()
        if K1: ref = ref + a1.float() @ b[:, K0:].float().t()
        err = ((c.float() - ref).abs().max() / ref.abs().max()).item()
    return ms, 2.0 * M * N * (K0 + K1) / ms / 1e9, err

def check_conv(variant, Bn, res, Cin, N, K1):
    """3x3 conv (+ optional 1x1 shortcut over K1 channels) against torch.nn.functional.conv2d (fp32)."""
    import torch.nn.functional as F
    x = torch.randn(Bn, Cin, res, res, device=dev)
    w = torch.randn(N, Cin, 3, 3, device=dev) * 0.05
    xb = x.to(torch.bfloat16); wb = w.to(torch.bfloat16)
    a = torch.zeros(Bn, res + 2, res + 2, Cin, dtype=torch.bfloat16, device=dev)
    a[:, 1:-1, 1:-1] = xb.permute(0, 2, 3, 1)
    # engine K order: 64-channel chunk outer, tap, channel
    wp = wb.reshape(N, Cin // 64, 64, 9).permute(0, 1, 3, 2).reshape(N, Cin * 9)
    ref = F.conv2d(xb.float(), wb.float(), padding=1)
    a1 = None
    if K1:
        s = torch.randn(Bn, K1, res, res, device=dev); w1 = torch.randn(N, K1, device=dev) * 0.05
        sb = s.to(torch.bfloat16); w1b = w1.to(torch.bfloat16)
        a1 = sb.permute(0, 2, 3, 1).contiguous().reshape(-1, K1)
        wp = torch.cat([wp, w1b], dim=1)
        ref = ref + torch.einsum("bkhw,nk->bnhw", sb.float(), w1b.float())
    wp = wp.contiguous()
    M = Bn * res * res
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    check(lib.natinf_debug_gemm(variant, M, N, Cin * 9, K1, 9, res.bit_length() - 1, 1, a.data_ptr(), a1.data_ptr() if K1 else None,
                                wp.data_ptr(), None, c.data_ptr(), 0, 1.0, 1, stream_ptr()), "debug_gemm")
    torch.cuda.synchronize()
    got = c.float().reshape(Bn, res, res, N).permute(0, 3, 1, 2)
    return ((got - ref).abs().max() / ref.abs().max()).item()


SHAPES = [  # (M, N, K0, K1, taps, res)  -- B=512 layer shapes, largest time first
    (524288, 128, 1152, 0, 9, 32), (524288, 128, 2304, 128, 9, 32), (131072, 256, 2304, 0, 9, 16), (131072, 256, 4608, 0, 9, 16),
    (524288, 256, 2304, 256, 9, 32), (32768, 256, 2304, 0, 9, 8), (8192, 256, 2304, 0, 9, 4), (131072, 256, 256, 0, 1, 0),
    (131072, 512, 256, 0, 1, 0),
]
if __name__ != "__main__":
    SHAPES = []
if len(sys.argv) > 1 and sys.argv[1] == "8ph":
    # correctness of the 256x256 variants on 256-aligned shapes (many repeats: the schedule is race-prone by nature), then speed
    vs = [16, 19, 20, 21, 22]
    for v in vs:
        for rep in range(1):
            errs = [run(v, 1024, 512, 256, 128, 1, 0, iters=2, check_ref=True)[2], run(v, 2048, 256, 1536, 0, 1, 0, iters=2, check_ref=True)[2],
                    run(v, 512, 768, 64, 0, 1, 0, iters=2, check_ref=True)[2], run(v, 256, 256, 128, 64, 1, 0, iters=2, check_ref=True)[2]]
            cerr = [check_conv(v, *c) for c in ((4, 16, 256, 256, 128), (8, 8, 128, 256, 64), (1, 32, 192, 256, 64), (2, 32, 64, 512, 0))]
            print(f"check {NAMES[v]:>12}: gemm " + " ".join(f"{e:.1e}" for e in errs) + " | conv " + " ".join(f"{e:.1e}" for e in cerr), flush=True)
    shapes = [(131072, 256, 2304, 0, 9, 16), (131072, 256, 4608, 0, 9, 16), (524288, 256, 2304, 256, 9, 32), (32768, 256, 2304, 0, 9, 8),
              (32768, 1536, 1536, 0, 1, 0), (32768, 6144, 1536, 0, 1, 0), (32768, 1536, 6144, 0, 1, 0), (32768, 3072, 1536, 0, 1, 0),
              (8192, 8192, 8192, 0, 1, 0), (4096, 4096, 4096, 0, 1, 0)]
    print(f"{'shape':>34} " + " ".join(f"{NAMES[v]:>12}" for v in vs))
    for (M, N, K0, K1, taps, res) in shapes:
        cells = [f"{run(v, M, N, K0, K1, taps, res, iters=10)[1]:7.0f}TF/s" for v in vs]
        print(f"{str((M, N, K0 + K1, taps)):>34} " + " ".join(f"{c:>12}" for c in cells), flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "sd3":
    vs = [int(v) for v in sys.argv[2:]] or [9, 10, 18, 16, 19, 13]
    shapes = [(32768, 1536, 1536, 0, 1, 0), (32768, 3072, 1536, 0, 1, 0), (32768, 6144, 1536, 0, 1, 0), (32768, 1536, 6144, 0, 1, 0),
              (65536, 1152, 1152, 0, 1, 0), (65536, 4608, 1152, 0, 1, 0), (65536, 1152, 4608, 0, 1, 0)]
    print(f"{'shape':>34} " + " ".join(f"{NAMES[v]:>13}" for v in vs))
    for (M, N, K0, K1, taps, res) in shapes:
        cells = [f"{run(v, M, N, K0, K1, taps, res, iters=10)[1]:7.0f}TF/s" for v in vs]
        print(f"{str((M, N, K0 + K1, taps)):>34} " + " ".join(f"{c:>13}" for c in cells), flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "one":
    # bench_gemm.py one <variant> <M> <N> <K0> <K1> <taps> <res> [iters]
    v, M, N, K0, K1, taps, res = map(int, sys.argv[2:9])
    it = int(sys.argv[9]) if len(sys.argv) > 9 else 5
    CMODE = int(sys.argv[10]) if len(sys.argv) > 10 else 0
    ms, tf, _ = run(v, M, N, K0, K1, taps, res, iters=it)
    print(f"{NAMES[v]} {(M, N, K0 + K1, taps)}: {ms*1e3:.1f} us  {tf:.0f} TF/s")
    sys.exit(0)
variants = ([int(v) for v in sys.argv[1:]] or [26, 27, 4, 6, 8]) if __name__ == "__main__" else []      # (imported by bench_blaslt.py: nothing runs; the defaults are shipped variants)
# correctness (plain GEMM with both K segments) for every variant first
for v in variants:
    ms, tf, err = run(v, 1000, 384, 256, 128, 1, 0, iters=2, check_ref=True)
    print(f"check {NAMES[v]:>12}: rel err {err:.2e}")
for v in variants:
    errs = []
    for (Bn, res, Cin, N, K1) in ((5, 32, 128, 128, 0), (3, 16, 256, 256, 128), (9, 8, 128, 256, 64), (2, 32, 192, 128, 64)):
        if v == 14 and res < 16:
            continue
        errs.append(check_conv(v, Bn, res, Cin, N, K1))
    print(f"conv check {NAMES[v]:>12}: " + " ".join(f"{e:.2e}" for e in errs))
if variants: print(f"{'shape':>34} " + " ".join(f"{NAMES[v]:>12}" for v in variants))
for (M, N, K0, K1, taps, res) in SHAPES:
    cells = []
    for v in variants:
        try:
            ms, tf, _ = run(v, M, N, K0, K1, taps, res)
            cells.append(f"{tf:7.0f}TF/s")
        except Exception as e:
            cells.append("       error")
    print(f"{str((M, N, K0 + K1, taps)):>34} " + " ".join(f"{c:>12}" for c in cells), flush=True)
