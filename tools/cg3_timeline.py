"""Where a k_conv_gn3 tile spends its shader clocks.  Needs a library built with EXTRA=-DNATINF_CG3_TIMELINE (NATINF_LIB=...: every launch then runs the
GroupNorm-partials epilogue; timings only).  usage: cg3_timeline.py res B cin N c1"""
import ctypes as C
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd._lib import lib, check, ptr, stream_ptr
res, B, cin, N, c1 = [int(v) for v in sys.argv[1:6]]
dev = "cuda"; M = B * res * res
x = torch.randn(B, res, res, cin, device=dev).bfloat16(); sc = torch.rand(B, cin, device=dev) + 0.5; sh = torch.randn(B, cin, device=dev) * 0.3
w = (torch.randn(N, 9 * cin + c1, device=dev) / (9 * cin) ** 0.5).bfloat16(); a1 = torch.randn(M, c1, device=dev).bfloat16() if c1 else None
bias = torch.randn(N, device=dev); out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); part = torch.zeros(M // 128, N // 4, 2, device=dev)
wf = torch.zeros_like(w)
ts = torch.zeros(128, dtype=torch.int64, device=dev)
f = lib.natinf_debug_cg3_timeline; f.restype = C.c_int; f.argtypes = [C.c_void_p]
check(f(ptr(ts)), "timeline")
check(lib.natinf_set_conv_gn_w128(7), "knob")
for sh in range(3):
    check(lib.natinf_set_conv_gn_w128_min_k(sh, 0), "min_k")
args = (res, B, N, cin, c1, ptr(x), ptr(sc), ptr(sh), ptr(w), ptr(wf), ptr(a1), ptr(bias), None, 0.7071, ptr(out), ptr(part))
for _ in range(3):
    check(lib.natinf_debug_conv_gn(*args, 1, stream_ptr()), "run"); torch.cuda.synchronize()
nh = cin // 32
for name, o in (("block 0", 0), ("block 300", 64)):
    t = ts[o:o + 64].tolist()
    if t[0] == 0: continue
    hcs = [t[4 + h + 1] - t[4 + h] for h in range(nh - 1)] + [t[40] - t[4 + nh - 1]]
    print(f"{name}: requests issued {t[1] - t[0]}, patch landed +{t[2] - t[1]}, normalisation of half-chunk 0 + first reads +{t[3] - t[2]} | half-chunks {hcs} = {sum(hcs) / (9 * nh):.0f} / tap | "
          f"shortcut segment {t[41] - t[40]}{' = %.0f / step' % ((t[41] - t[40]) / (c1 // 32)) if c1 else ''} | drain + args {t[42] - t[41]} | epilogue {t[43] - t[42]} | whole tile {t[43] - t[0]}")
