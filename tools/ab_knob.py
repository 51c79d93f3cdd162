"""Same-box A/B of a tuning knob on the NCSN++ forward at B=512: ab_knob.py <abi function name> [values...]  (default 0 1)"""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd._lib import lib, check
from naturaldiffusion_amd.ncsnpp import NCSNppEngine
from naturaldiffusion_amd.synth import synthetic_flat_params
fn = getattr(lib, sys.argv[1]); vals = [int(v) for v in sys.argv[2:]] or [0, 1]
eng = NCSNppEngine(synthetic_flat_params(0), max_batch=512)
x = torch.randn(512, 3, 32, 32, device="cuda"); t = torch.rand(512, device="cuda") * 999
outs = {}
for rep in range(3):
    for v in vals:
        check(fn(v), "set")
        for _ in range(2): outs[v] = eng(x, t)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): eng(x, t)
        torch.cuda.synchronize()
        print(f"{sys.argv[1]}({v}): {(time.perf_counter() - t0) * 100:.2f} ms per forward", flush=True)
print("max rel diff between settings:", ((outs[vals[0]] - outs[vals[-1]]).abs().max() / outs[vals[0]].abs().max()).item())
