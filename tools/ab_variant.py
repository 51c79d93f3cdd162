"""Time the NCSN++ forward at B=512 with a forced GEMM variant (0 = auto): python tools/ab_variant.py 0 15 14 0"""
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
for v in [int(a) for a in sys.argv[1:]]:
    check(lib.natinf_set_gemm_variant(v), "set")
    for _ in range(2): eng(x, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): eng(x, t)
    torch.cuda.synchronize()
    print(f"variant {v}: {(time.perf_counter() - t0) * 100:.2f} ms per forward")
check(lib.natinf_set_gemm_variant(0), "set")
