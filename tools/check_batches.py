"""Default plan against the plan with every fusion knob off, on odd batch sizes (tails of the multi-image tiles, the attention block pairing, partial tables):
check_batches.py [B ...]"""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd._lib import lib, check
from naturaldiffusion_amd.ncsnpp import NCSNppEngine
from naturaldiffusion_amd.synth import synthetic_flat_params
Bs = [int(a) for a in sys.argv[1:]] or [1, 3, 5, 7, 9, 13, 33]
knobs = [lib.natinf_set_fuse_gn8, lib.natinf_set_fuse_gn4, lib.natinf_set_fuse_fin, lib.natinf_set_attn_qkv, lib.natinf_set_attn_proj, lib.natinf_set_fuse_head]
worst = 0.0
for arch in ("ncsnpp", "ddpm"):
    flat = synthetic_flat_params(0) if arch == "ncsnpp" else None
    if flat is None:
        from naturaldiffusion_amd.synth import synthetic_state_dict
        from naturaldiffusion_amd.ncsnpp import param_layout
        g = torch.Generator().manual_seed(5)
        flat = torch.cat([(torch.randn(*s, generator=g) * (0.02 if len(s) > 1 else 0.1) + (1.0 if n.endswith("GroupNorm_0.weight") or n.endswith("GroupNorm_1.weight") else 0.0)).reshape(-1)
                          for n, s in param_layout(128, arch)])
    for B in Bs:
        x = torch.randn(B, 3, 32, 32, device="cuda"); t = torch.rand(B, device="cuda") * 999
        eng = NCSNppEngine(flat, max_batch=B, arch=arch)
        y = eng(x, t).clone()
        for k in knobs: check(k(0), "set")
        try:
            ref = NCSNppEngine(flat, max_batch=B, arch=arch)(x, t).clone()
        finally:
            for k in knobs: k(1)
        torch.cuda.synchronize()
        e = ((y - ref).abs().max() / ref.abs().max()).item()
        worst = max(worst, e)
        print(f"{arch} B={B}: max rel diff fused vs unfused {e:.3e}  finite={bool(torch.isfinite(y).all())}", flush=True)
        assert torch.isfinite(y).all() and e < 3e-2
print("worst", worst)
