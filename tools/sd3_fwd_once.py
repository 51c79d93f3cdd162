"""Two SD3-size MMDiT forwards of 8 sequences (for rocprofv3 --pmc passes: tools/profile_sd3.sh).  usage: sd3_fwd_once.py [fp8]"""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd.mmdit import MMDiTEngine, SD3_MEDIUM
from naturaldiffusion_amd.synth import synthetic_mmdit_flat
fp8 = "fp8" in sys.argv
cfg = dict(SD3_MEDIUM)
eng = MMDiTEngine(synthetic_mmdit_flat(64, seed=0, **cfg), max_batch=8, grid=64, ctx_tokens=333, fp8=fp8, **cfg)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(8, 16, 128, 128, device="cuda", generator=g); t = torch.rand(8, device="cuda", generator=g) * 1000
e = torch.randn(8, 333, cfg["joint_dim"], device="cuda", generator=g); p = torch.randn(8, cfg["pooled_dim"], device="cuda", generator=g)
for _ in range(2):
    o = eng.forward(x, t, e, p)
torch.cuda.synchronize()
print("ok", float(o.float().abs().mean()))
