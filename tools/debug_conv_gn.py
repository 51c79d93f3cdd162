import sys
from pathlib import Path
import numpy as np, torch, torch.nn.functional as F
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from naturaldiffusion_amd._lib import lib, check, ptr, stream_ptr
from test_gpu_conv_gn import _pack
res, B, cin, N, c1 = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (32, 2, 128, 128, 0)
g = torch.Generator().manual_seed(1)
bf = lambda t: t.bfloat16().float()
x = bf(torch.randn(B, res, res, cin, generator=g))
scale = torch.rand(B, cin, generator=g) * 1.5 + 0.25; shift = torch.randn(B, cin, generator=g) * 0.5
w = bf(torch.randn(N, cin, 3, 3, generator=g) / np.sqrt(9 * cin))
bias = torch.randn(N, generator=g) * 0.1
h = bf(F.silu(x * scale[:, None, None, :] + shift[:, None, None, :]))
ref = (F.conv2d(h.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1).reshape(B * res * res, N) + bias.double()).float()
dev = "cuda"
xd, wd, scd, shd, bd = x.bfloat16().to(dev).contiguous(), _pack(w * -0.6931471805599453, None).bfloat16().to(dev), scale.to(dev), shift.to(dev), bias.to(dev)
wf = torch.zeros_like(wd)
out = torch.zeros(B * res * res, N, dtype=torch.bfloat16, device=dev)
for parts in (False, True):
    part = torch.zeros(B * res * res // 256, N // 4, 2, device=dev) if parts else None
    check(lib.natinf_debug_conv_gn(res, B, N, cin, c1, ptr(xd), ptr(scd), ptr(shd), ptr(wd), ptr(wf), None, ptr(bd), None, 1.0, ptr(out), ptr(part), 1, stream_ptr()), "x")
    torch.cuda.synchronize()
    got = out.float().cpu()
    d = (got - ref).abs()
    print("parts", parts, "max|got|", got.abs().max().item(), "max|ref|", ref.abs().max().item(), "max err", d.max().item(), "frac bad", (d > 0.05).float().mean().item())
    bad = (d > 0.05).nonzero()
    if len(bad):
        rows = bad[:, 0]; cols = bad[:, 1]
        print(" bad rows (first 20):", rows[:20].tolist(), " bad cols uniq:", cols.unique()[:40].tolist(), " rows uniq count", rows.unique().numel())
        r0 = rows.unique()
        print(" bad row pattern: y =", ((r0 % (res * res)) // res).unique()[:40].tolist(), " x =", (r0 % res).unique()[:40].tolist(), "img", (r0 // (res * res)).unique().tolist())
