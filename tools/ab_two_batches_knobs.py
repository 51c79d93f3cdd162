"""Plan knobs re-measured with two batches in flight (two engine handles on two streams): choices made for one stream (fill the chip) need not hold when the
other lane fills the gaps.  ms per 512 images, forward only."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd._lib import lib
from naturaldiffusion_amd.ncsnpp import NCSNppEngine
from naturaldiffusion_amd.synth import synthetic_flat_params
p = synthetic_flat_params(0)
xa = torch.randn(512, 3, 32, 32, device="cuda"); xb = torch.randn(512, 3, 32, 32, device="cuda"); t = torch.rand(512, device="cuda") * 999
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def measure(tag):
    ea, eb = NCSNppEngine(p, max_batch=512), NCSNppEngine(p, max_batch=512)       # the plan is built at the first forward with the knobs of that moment
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
    print(f"{tag:34s}: one stream {res[0]:.3f} / {res[2]:.3f}   two streams {res[1]:.3f} / {res[3]:.3f} ms per 512 images", flush=True)
    del ea, eb
measure("default")
for name, setter, off, on in (("conv_gn_warm 0", lib.natinf_set_conv_gn_warm, 0, 15), ("fuse_gn4 0 (4x4 on split-K GEMMs)", lib.natinf_set_fuse_gn4, 0, 1),
                              ("fuse_fin 0", lib.natinf_set_fuse_fin, 0, 1), ("attn_waves8 0", lib.natinf_set_attn_waves8, 0, 1),
                              ("attn_qkv 0", lib.natinf_set_attn_qkv, 0, 1), ("conv_gn_wide 1 (16x16 only)", lib.natinf_set_conv_gn_wide, 1, 3),
                              ("fuse_gn8 0", lib.natinf_set_fuse_gn8, 0, 1)):
    setter(off)
    try: measure(name)
    finally: setter(on)
measure("default again")
