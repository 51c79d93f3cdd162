"""SD3-size MMDiT forward (8 sequences) with the text stream's launches on the caller's stream (0), on a second HIP stream (1):
a fresh engine per setting (the stream is created at the engine's first forward).  usage: ab_sd3_text_stream.py [fp8] (GPU box)"""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd._lib import lib, check
from naturaldiffusion_amd.mmdit import MMDiTEngine, SD3_MEDIUM
from naturaldiffusion_amd.synth import synthetic_mmdit_flat
fp8 = "fp8" in sys.argv
cfg = dict(SD3_MEDIUM)
flat = synthetic_mmdit_flat(64, seed=0, **cfg)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(8, 16, 128, 128, device="cuda", generator=g); t = torch.rand(8, device="cuda", generator=g) * 1000
e = torch.randn(8, 333, cfg["joint_dim"], device="cuda", generator=g); p = torch.randn(8, cfg["pooled_dim"], device="cuda", generator=g)
for rep in range(2):
    for v in (0, 1):
        check(lib.natinf_set_mmdit_text_stream(v), "set")
        eng = MMDiTEngine(flat, max_batch=8, grid=64, ctx_tokens=333, fp8=fp8, **cfg)
        for _ in range(2): eng.forward(x, t, e, p)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): eng.forward(x, t, e, p)
        torch.cuda.synchronize()
        print(f"text_stream({v}){' fp8' if fp8 else ''}: {(time.perf_counter() - t0) * 200:.2f} ms per forward of 8 sequences", flush=True)
        del eng
