"""Same-box A/B of a plan-build-time knob (read by natinf_ncsnpp_create): two engines in one process.
usage: ab_build_knob.py <abi function name> [values...] [B=512]  (default values 0 1)"""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd._lib import lib, check
from naturaldiffusion_amd.ncsnpp import NCSNppEngine
from naturaldiffusion_amd.synth import synthetic_flat_params
B = next((int(a[2:]) for a in sys.argv if a.startswith("B=")), 512)
fn = getattr(lib, sys.argv[1]); vals = [int(v) for v in sys.argv[2:] if not v.startswith("B=")] or [0, 1]
flat = synthetic_flat_params(0)
engs = {}
for v in vals:
    check(fn(v), "set"); engs[v] = NCSNppEngine(flat, max_batch=B)
x = torch.randn(B, 3, 32, 32, device="cuda"); t = torch.rand(B, device="cuda") * 999
outs = {}
for rep in range(3):
    for v in vals:
        for _ in range(2): outs[v] = engs[v](x, t)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): engs[v](x, t)
        torch.cuda.synchronize()
        print(f"{sys.argv[1]}({v}) B={B}: {(time.perf_counter() - t0) * 100:.2f} ms per forward", flush=True)
print("outputs identical:", torch.equal(outs[vals[0]], outs[vals[-1]]), " max rel diff:", ((outs[vals[0]] - outs[vals[-1]]).abs().max() / outs[vals[0]].abs().max()).item())
