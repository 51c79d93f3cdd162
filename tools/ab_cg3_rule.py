"""Which K threshold per shape should send a fused convolution to k_conv_gn3?  NCSN++ forward at B = 512 under several natinf_set_conv_gn_w128_min_k settings
(same process, interleaved rounds): ab_cg3_rule.py"""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd._lib import lib, check
from naturaldiffusion_amd.ncsnpp import NCSNppEngine
from naturaldiffusion_amd.synth import synthetic_flat_params
eng = NCSNppEngine(synthetic_flat_params(0), max_batch=512)
x = torch.randn(512, 3, 32, 32, device="cuda"); t = torch.rand(512, device="cuda") * 999
RULES = {"off": None, "default 2304/0/2560": (2304, 0, 2560), "all K": (0, 0, 0), "1408/0/2304": (1408, 0, 2304), "2304/0/2304": (2304, 0, 2304), "3456/0/2816": (3456, 0, 2816),
         "2304/9999/2560": (2304, 99999, 2560), "99999/0/99999": (99999, 0, 99999), "1152+/0/1152+": (1152, 0, 1152)}
best = {k: 1e9 for k in RULES}
for rep in range(3):
    for name, r in RULES.items():
        check(lib.natinf_set_conv_gn_w128(0 if r is None else 7), "mask")
        if r is not None:
            for sh, k in enumerate(r): check(lib.natinf_set_conv_gn_w128_min_k(sh, k), "min_k")
        for _ in range(2): eng(x, t)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): eng(x, t)
        torch.cuda.synchronize()
        best[name] = min(best[name], (time.perf_counter() - t0) * 100)
for name, v in best.items(): print(f"{name:24s} {v:.2f} ms per forward", flush=True)
