"""One forward of the engine on seeded inputs, output and taps saved -- run once per library build (NATINF_LIB) and compared bitwise:
wait states added around the hand-written memory instructions (-DNATINF_ASM_PAD) must not change a single bit."""
import sys, torch
sys.path.insert(0, "/root/repo")
from naturaldiffusion_amd.ncsnpp import NCSNppEngine, module_table
from naturaldiffusion_amd.synth import synthetic_flat_params
tag = sys.argv[1]
B = 64
flat = synthetic_flat_params(0)
eng = NCSNppEngine(flat, max_batch=B, keep_activations=True)
g = torch.Generator(device="cpu").manual_seed(5)
x = torch.randn(B, 3, 32, 32, generator=g).cuda(); t = (torch.rand(B, generator=g) * 999).cuda()
mods = module_table()
def shape_of(m):
    idx, kind, cin, cout, up, down, res, _ = m
    ro = res * 2 if up else (res // 2 if down else res)
    if kind in ("res", "attn"): return (B, ro, ro, cout)
    if kind == "conv" and cin == 3: return (B, res, res, cout)
    return None
outs = []
for rep in range(3):
    o = eng(x, t).clone(); torch.cuda.synchronize()
    outs.append(o)
print(tag, "self-consistent over 3 runs:", all(torch.equal(outs[0], o) for o in outs[1:]))
taps = {}
for m in mods:
    s = shape_of(m)
    if s is not None:
        try: taps[m[0]] = eng.tap(m[0], s).cpu()
        except Exception: pass
torch.save({"out": outs[0].cpu(), "taps": taps}, f"/tmp/pad_{tag}.pt")
