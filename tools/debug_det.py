"""Two engines on two HIP streams (the victim keeps its activations): per-module taps against the solo run -- which module differs first, in how many
elements and pixels of which image.  (First tool of the hunt in DESIGN.md section 5; with the fixed dpp_row_sum it prints nothing.)"""
import sys, torch
sys.path.insert(0, "/root/repo")
from naturaldiffusion_amd._lib import lib
from naturaldiffusion_amd.ncsnpp import NCSNppEngine, module_table
from naturaldiffusion_amd.synth import synthetic_flat_params
flat = synthetic_flat_params(0)
B = 64
mods = module_table()
def shape_of(m):
    idx, kind, cin, cout, up, down, res, _ = m
    ro = res * 2 if up else (res // 2 if down else res)
    if kind in ("res", "attn"): return (B, ro, ro, cout)
    if kind == "conv" and cin == 3: return (B, res, res, cout)
    return None
eA = NCSNppEngine(flat, max_batch=B, keep_activations=True)
lib.natinf_set_fuse_gn(0); eU = NCSNppEngine(flat, max_batch=B); lib.natinf_set_fuse_gn(1)
x = torch.randn(B, 3, 32, 32, device="cuda"); x2 = torch.randn(B, 3, 32, 32, device="cuda"); t = torch.rand(B, device="cuda") * 999
def taps(eng):
    out = {}
    for m in mods:
        s = shape_of(m)
        if s is not None:
            try: out[m[0]] = eng.tap(m[0], s).clone()
            except Exception: pass
    return out
eA(x, t); torch.cuda.synchronize(); ref = taps(eA)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
shown = 0
for it in range(60):
    with torch.cuda.stream(sb): eU(x2, t)
    with torch.cuda.stream(sa): eA(x, t)
    torch.cuda.synchronize()
    cur = taps(eA)
    bad = [k for k in sorted(cur) if not torch.equal(cur[k], ref[k])]
    if bad and shown < 5:
        k = bad[0]; d = (cur[k] - ref[k]).abs()
        b0 = int(d.amax(dim=(1, 2, 3)).argmax()); di = d[b0]; r = ref[k][b0].abs()
        n = int((di > 0).sum()); tot = di.numel()
        rel = (di / r.clamp_min(1e-6))[di > 0]
        pix = (di.amax(dim=2) > 0).sum().item()
        hot = di.sum(dim=2); hy, hx = divmod(int(hot.argmax()), hot.shape[1])
        print(f"it {it}: module {mods[k][:7]} image {b0}: {n}/{tot} elements differ ({100.0 * n / tot:.2f} %), in {pix}/{hot.numel()} pixels; |d|/|x| median {float(rel.median()):.4f} max {float(rel.max()):.4f}; hottest pixel ({hy},{hx}) sum|d| {float(hot.max()):.3f} vs mean {float(hot.mean()):.4f}", flush=True)
        shown += 1
