"""Per-module errors of the NCSN++ engine against the fp32 oracle, by plan, at B = 512 (the two golden samples in slots 0-1 of a batch of random neighbours; keep_activations: 14.7 GB)."""
import sys, json
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd._lib import lib, check
from naturaldiffusion_amd.ncsnpp import NCSNppEngine, flatten_state_dict
from oracle import ncsnpp_oracle as N
params = N.make_params(seed=0); flat = flatten_state_dict(params)
fx = np.load(ROOT / "tests/golden/ncsnpp_forward.npz")
gx, gl = torch.from_numpy(fx["x"]), torch.from_numpy(fx["labels"])
taps = {}; y_ref = N.forward(params, gx, gl, taps)
g = torch.Generator().manual_seed(12)
x = torch.randn(512, 3, 32, 32, generator=g); labels = torch.rand(512, generator=g) * 999
x[0:2] = gx; labels[0:2] = gl
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
prev = None
for name, blk, mask, mk in (("round-4 plan", 0, 0, (2304, 0, 2304)), ("+ attention block", 1, 0, (2304, 0, 2304)), ("+ k_conv_gn3 (2304/0/2560)", 0, 7, (2304, 0, 2560)),
                            ("+ k_conv_gn3 (2304/0/2304)", 0, 7, (2304, 0, 2304)), ("both (round-5 default)", 1, 7, (2304, 0, 2304)),
                            ("k_attn_blk256_v2 + k_conv_gn3", 2, 7, (2304, 0, 2304))):
    check(lib.natinf_set_attn_block(blk), "blk"); check(lib.natinf_set_conv_gn_w128(mask), "mask")
    for s, k in enumerate(mk): check(lib.natinf_set_conv_gn_w128_min_k(s, k), "mk")
    eng = NCSNppEngine(flat, max_batch=512, keep_activations=True)
    y = eng(x.cuda(), labels.cuda()); torch.cuda.synchronize()
    e = {}
    for k in range(2, 53):
        full = eng.tap(k, (512,) + tuple(taps[k].shape[1:])); e[k] = rel(full[0:2].cpu(), taps[k]); del full
    print(f"{name:32s} y {rel(y[0:2].cpu(), y_ref):.4f} | 8-16: " + " ".join(f"{e[k]:.4f}" for k in range(8, 17)) + f" | max 41-47 {max(e[k] for k in range(41, 48)):.4f} | max 48-52 {max(e[k] for k in range(48, 53)):.4f} | max 22-34 {max(e[k] for k in range(22, 35)):.4f}", flush=True)
    del eng; torch.cuda.empty_cache()
lib.natinf_set_attn_block(-1); lib.natinf_set_conv_gn_w128(7)
