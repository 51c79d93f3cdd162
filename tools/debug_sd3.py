import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from naturaldiffusion_amd.mmdit import MMDiTEngine, SD3_MEDIUM
from naturaldiffusion_amd.synth import synthetic_mmdit_flat
dev = torch.device('cuda:0')
tc = 333
def run(cfg, grid, B, fill, tag):
    flat = synthetic_mmdit_flat(grid=grid, seed=0, **cfg)
    eng = MMDiTEngine(flat, max_batch=B, grid=grid, ctx_tokens=tc, device=dev, **cfg)
    g = torch.Generator(device=dev).manual_seed(10)
    z = torch.randn(B, 16, 2 * grid, 2 * grid, device=dev, generator=g)
    text = torch.randn(B, tc, cfg["joint_dim"], device=dev, generator=g)
    pooled = torch.randn(B, cfg["pooled_dim"], device=dev, generator=g)
    t = torch.full((B,), 500.0, device=dev)
    for f in fill:
        if f == "nan": eng._ws.view(torch.int16).fill_(0x7fc0)          # bf16 NaN pattern / fp32 NaN-ish
        elif f == "zero": eng._ws.zero_()
        o = eng.forward(z, t, text, pooled)
        print(tag, f, "finite", bool(torch.isfinite(o).all()), "absmax", float(o.abs().nan_to_num(0).max()), "nan frac", float(torch.isnan(o).float().mean()), flush=True)
    del eng
run(SD3_MEDIUM, 64, 8, ["zero", "nan", "zero"], "medium B8")
run(SD3_MEDIUM, 64, 2, ["zero", "nan"], "medium B2")
run(dict(layers=2, heads=24, joint_dim=4096, pooled_dim=2048, in_ch=16), 64, 2, ["zero", "nan"], "L2 wide")
run(dict(layers=2, heads=2, joint_dim=64, pooled_dim=32, in_ch=16), 64, 2, ["zero", "nan"], "narrow")
