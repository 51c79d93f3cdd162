import sys
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from naturaldiffusion_amd._lib import lib, check
from naturaldiffusion_amd.ncsnpp import NCSNppEngine
from naturaldiffusion_amd.synth import synthetic_flat_params
flat = synthetic_flat_params(0)
x = torch.randn(512, 3, 32, 32, device="cuda"); t = torch.rand(512, device="cuda") * 999
for on in (1, 0):
    check(lib.natinf_set_attn_block(on), "k")
    eng = NCSNppEngine(flat, max_batch=512)
    for _ in range(5): eng(x, t)
    torch.cuda.synchronize(); del eng
