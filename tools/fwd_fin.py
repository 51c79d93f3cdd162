"""NCSN++ forwards at B=512 for rocprofv3 traces with a given natinf_set_fuse_fin plan: fwd_fin.py <mode 0..3> [n]"""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd._lib import lib, check
from naturaldiffusion_amd.ncsnpp import NCSNppEngine
from naturaldiffusion_amd.synth import synthetic_flat_params
check(lib.natinf_set_fuse_fin(int(sys.argv[1])), "set")
eng = NCSNppEngine(synthetic_flat_params(0), max_batch=512)
x = torch.randn(512, 3, 32, 32, device="cuda"); t = torch.rand(512, device="cuda") * 999
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10): eng(x, t)
torch.cuda.synchronize()
