"""Inception-V3 pool3 engine alone (GPU box): images/s at the FID batch (500) and at the reference's 50; run under
`rocprofv3 --kernel-trace --stats -- python3 tools/bench_inception.py 500 3` for the per-kernel split (profiles/r05/inception_*)."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd.inception import InceptionEngine
from naturaldiffusion_amd.synth import synthetic_inception_flat

B = int(sys.argv[1]) if len(sys.argv) > 1 else 500
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
eng = InceptionEngine(synthetic_inception_flat(0), max_batch=B, device=dev)
g = torch.Generator(device="cpu").manual_seed(1)
x = torch.randint(0, 256, (B, 32, 32, 3), dtype=torch.uint8, generator=g).to(dev)
f = eng(x); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    f = eng(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print({"B": B, "ms": round(dt * 1e3, 2), "images_per_s": round(B / dt, 1), "tflops_2mac_one_term": round(11.42e9 * B / dt / 1e12, 1),
       "feat_mean": float(f.mean()), "feat_absmax": float(f.abs().max())}, flush=True)
