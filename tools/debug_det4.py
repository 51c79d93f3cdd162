"""A libnatinf engine on one HIP stream with NaN-poisoning neighbour kernels (tools/probe/poison.hip) on another: a kernel
that consumes registers or LDS it has not written produces NaN (or garbage) instead of a slightly different number."""
import ctypes, sys, torch
sys.path.insert(0, "/root/repo")
from naturaldiffusion_amd.ncsnpp import NCSNppEngine, module_table
from naturaldiffusion_amd.synth import synthetic_flat_params
P = ctypes.CDLL("/root/repo/tools/probe/libpoison.so")
P.poison_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
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
x = torch.randn(B, 3, 32, 32, device="cuda"); t = torch.rand(B, device="cuda") * 999
def taps(eng):
    out = {}
    for m in mods:
        s = shape_of(m)
        if s is not None:
            try: out[m[0]] = eng.tap(m[0], s).clone()
            except Exception: pass
    return out
ref_o = eA(x, t).clone(); torch.cuda.synchronize(); ref = taps(eA)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for name, (nr, nl, pat, ldsb, spin) in {
    "idle neighbour": (0, 0, 0, 0, 0),
    "registers = NaN": (2048, 0, 0x7fc07fc0, 0, 60000),
    "registers = 1.0f": (2048, 0, 0x3f800000, 0, 60000),
    "registers = 0": (2048, 0, 0, 0, 60000),
    "LDS 64K = NaN": (0, 1024, 0x7fc07fc0, 65536, 0),
    "LDS 16K = NaN": (0, 2048, 0x7fc07fc0, 16384, 0),
    "LDS 16K = 0": (0, 2048, 0, 16384, 0),
    "both NaN": (2048, 1024, 0x7fc07fc0, 32768, 60000),
}.items():
    nbad = nnan = 0; first = None
    for it in range(20):
        with torch.cuda.stream(sb):
            for _ in range(150):
                rc = P.poison_launch(sb.cuda_stream, nr, nl, pat, ldsb, spin); assert rc == 0, rc
        with torch.cuda.stream(sa): o = eA(x, t)
        torch.cuda.synchronize()
        if not torch.equal(o, ref_o):
            nbad += 1; nnan += int(not bool(torch.isfinite(o).all()))
            if first is None:
                cur = taps(eA)
                bad = [k for k in sorted(cur) if not torch.equal(cur[k], ref[k])]
                if bad:
                    k = bad[0]; d = (cur[k].float() - ref[k].float())
                    first = f"first module {mods[k][:7]}, nan {int(torch.isnan(cur[k]).sum())}, max|d| {float(d.nan_to_num(0).abs().max()):.4f}"
    print(f"{name:18s}: {nbad:2d}/20 differ, {nnan} with non-finite values; {first}", flush=True)
