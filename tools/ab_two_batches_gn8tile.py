"""The 8x8 level's tile re-measured with two batches in flight (a -DNATINF_DEV library: NATINF_LIB=...): 64-pixel x 256-channel tiles (one image per tile, 512 blocks, two per
CU; the shipped form) against 128 x 256 (two images per tile, 256 blocks, one per CU and half the L2 -> register weight traffic) -- the second lane's launch can take the
CU's other slot.  ms per 512 images, forward only, one stream / two streams."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd._lib import lib, check
from naturaldiffusion_amd.ncsnpp import NCSNppEngine
from naturaldiffusion_amd.synth import synthetic_flat_params
p = synthetic_flat_params(0)
xa = torch.randn(512, 3, 32, 32, device="cuda"); xb = torch.randn(512, 3, 32, 32, device="cuda"); t = torch.rand(512, device="cuda") * 999
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def measure(tag):
    ea = NCSNppEngine(p, max_batch=512); eb = ea.clone()
    def par(n):
        for _ in range(n):
            with torch.cuda.stream(sa): ea(xa, t)
            with torch.cuda.stream(sb): eb(xb, t)
    def seq(n):
        for _ in range(n): ea(xa, t); eb(xb, t)
    res = []
    for fn in (seq, par, seq, par):
        fn(2); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(8); torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 16 * 1e3)
    print(f"{tag:40s}: one stream {res[0]:.3f} / {res[2]:.3f}   two streams {res[1]:.3f} / {res[3]:.3f} ms per 512 images", flush=True)
for rep in range(2):
    for one_image in (1, 0):
        check(lib.natinf_set_conv_gn8_tile(one_image), "natinf_set_conv_gn8_tile (needs a -DNATINF_DEV library)")
        measure("64 x 256 tiles (one image)" if one_image else "128 x 256 tiles (two images)")
check(lib.natinf_set_conv_gn8_tile(1), "reset")
