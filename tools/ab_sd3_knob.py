"""Same-box A/B of a tuning knob on the SD3-size MMDiT forward (8 sequences): ab_sd3_knob.py <abi function> [values...] [fp8]"""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd._lib import lib, check
from naturaldiffusion_amd.mmdit import MMDiTEngine, SD3_MEDIUM
from naturaldiffusion_amd.synth import synthetic_mmdit_flat
fp8 = "fp8" in sys.argv
fn = getattr(lib, sys.argv[1]); vals = [int(v) for v in sys.argv[2:] if v != "fp8"] or [0, 1]
cfg = dict(SD3_MEDIUM)
eng = MMDiTEngine(synthetic_mmdit_flat(64, seed=0, **cfg), max_batch=8, grid=64, ctx_tokens=333, fp8=fp8, **cfg)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(8, 16, 128, 128, device="cuda", generator=g); t = torch.rand(8, device="cuda", generator=g) * 1000
e = torch.randn(8, 333, cfg["joint_dim"], device="cuda", generator=g); p = torch.randn(8, cfg["pooled_dim"], device="cuda", generator=g)
for rep in range(2):
    for v in vals:
        check(fn(v), "set")
        for _ in range(2): eng.forward(x, t, e, p)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): eng.forward(x, t, e, p)
        torch.cuda.synchronize()
        print(f"{sys.argv[1]}({v}){' fp8' if fp8 else ''}: {(time.perf_counter() - t0) * 200:.2f} ms per forward of 8 sequences", flush=True)
