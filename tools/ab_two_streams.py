"""One NCSN++ forward at B = 512 on one stream against two forwards at B = 256 on two streams (do the under-occupied low-resolution
kernels of one half fill in under the other half's convolutions?): ms per 512 images."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd.ncsnpp import NCSNppEngine
from naturaldiffusion_amd.synth import synthetic_flat_params
p = synthetic_flat_params(0)
e512 = NCSNppEngine(p, max_batch=512)
ea, eb = NCSNppEngine(p, max_batch=256), NCSNppEngine(p, max_batch=256)
x = torch.randn(512, 3, 32, 32, device="cuda"); t = torch.rand(512, device="cuda") * 999
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def one(n):
    for _ in range(n): e512(x, t)
def two(n):
    for _ in range(n):
        with torch.cuda.stream(sa): ea(x[:256], t[:256])
        with torch.cuda.stream(sb): eb(x[256:], t[256:])
for rep in range(3):
    for name, fn in (("1 x 512", one), ("2 x 256", two)):
        fn(2); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(10); torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) * 100:.2f} ms per 512 images", flush=True)
