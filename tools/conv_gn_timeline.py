"""Where a k_conv_gn tile spends its shader clocks (needs a build with EXTRA="-DNATINF_DEV -DNATINF_CG_TIMELINE": NATINF_LIB=...; that build spills 2-4 vector registers in the 256-register instantiations of k_conv_gn2 and can fault -- see csrc/conv_gn.h).
usage: conv_gn_timeline.py res B cin N c1"""
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
bias = torch.randn(N, device=dev); out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); part = torch.zeros(M // min(64, res * res), N // 4, 2, device=dev)
wf = torch.zeros_like(w)
ts = torch.zeros(24, dtype=torch.int64, device=dev)
check(lib.natinf_debug_timestamps(ptr(ts)), "ts")
args = (res, B, N, cin, c1, ptr(x), ptr(sc), ptr(sh), ptr(w), ptr(wf), ptr(a1), ptr(bias), None, 0.7071, ptr(out), ptr(part))
for _ in range(3):
    check(lib.natinf_debug_conv_gn(*args, 1, stream_ptr()), "run"); torch.cuda.synchronize()
for blk, o in (("block 0", 8), ("block 777", 16)):
    t = ts[o:o + 8].tolist(); nk = max(t[6], 1)
    print(f"{blk}: prologue {t[0]}  loop {t[4]} = {t[4] / nk:.0f}/tap over {nk} taps [wait+barrier {t[1] / nk:.0f}  dma+norm {t[2] / nk:.0f}  mfma section {t[3] / nk:.0f}]  epilogue {t[5]}")
e = ts[:8].tolist()
print(f"block 0 prologue: index math + requests {e[5]}, wait for the patch {e[6]}, normalisation of half-chunk 0 {ts[8].item() - e[5] - e[6]}")
print(f"block 0 epilogue: until its start stamp {e[2] - e[7]}, register phase + slab {e[3] - e[2]}, copy-out {e[4] - e[3]} (then the GroupNorm partial reduction)")
