"""Two independent batches of 512 on two streams (two engine handles) against the same two batches one after the other on one stream: ms per 512 images."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd.ncsnpp import NCSNppEngine
from naturaldiffusion_amd.synth import synthetic_flat_params
p = synthetic_flat_params(0)
ea, eb = NCSNppEngine(p, max_batch=512), NCSNppEngine(p, max_batch=512)
xa = torch.randn(512, 3, 32, 32, device="cuda"); xb = torch.randn(512, 3, 32, 32, device="cuda"); t = torch.rand(512, device="cuda") * 999
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def seq(n):
    for _ in range(n): ea(xa, t); eb(xb, t)
def par(n):
    for _ in range(n):
        with torch.cuda.stream(sa): ea(xa, t)
        with torch.cuda.stream(sb): eb(xb, t)
for rep in range(3):
    for name, fn in (("sequential", seq), ("two streams", par)):
        fn(2); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(6); torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) / 12 * 1e3:.2f} ms per 512 images", flush=True)
