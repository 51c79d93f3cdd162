"""Per-module errors of the NCSN++ engine against the fp32 oracle (the two golden samples at B = 2) under several plans: which switch moves which module.
tap_errors_by_plan.py"""
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
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
for name, blk, mask, mk in (("round-4 plan", 0, 0, (2304, 0, 2304)), ("+ attention block", 1, 0, (2304, 0, 2304)), ("+ k_conv_gn3 (2304/0/2560)", 0, 7, (2304, 0, 2560)),
                            ("+ k_conv_gn3 (2304/0/2304)", 0, 7, (2304, 0, 2304)), ("both (default)", 1, 7, (2304, 0, 2304)), ("both, every K on k_conv_gn3", 1, 7, (0, 0, 0))):
    check(lib.natinf_set_attn_block(blk), "blk"); check(lib.natinf_set_conv_gn_w128(mask), "mask")
    for s, k in enumerate(mk): check(lib.natinf_set_conv_gn_w128_min_k(s, k), "mk")
    eng = NCSNppEngine(flat, max_batch=2, keep_activations=True)
    y = eng(gx.cuda(), gl.cuda()); torch.cuda.synchronize()
    e = {k: rel(eng.tap(k, tuple(taps[k].shape)).cpu(), taps[k]) for k in range(2, 53)}
    print(f"{name:32s} y {rel(y.cpu(), y_ref):.4f} | 8-16: " + " ".join(f"{e[k]:.4f}" for k in range(8, 17)) + f" | max 41-47 {max(e[k] for k in range(41, 48)):.4f} | max 48-52 {max(e[k] for k in range(48, 53)):.4f} | max 22-34 {max(e[k] for k in range(22, 35)):.4f}", flush=True)
    del eng
lib.natinf_set_attn_block(1); lib.natinf_set_conv_gn_w128(7)
