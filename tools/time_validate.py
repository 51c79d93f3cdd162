"""Where a ValidateNaturalInference.natural_inference("ddim", 24) call of 8 images goes (GPU box): wall time of the set-up, the 24-step loop, the decode and the
PNG row, each closed by a device synchronisation, next to the un-instrumented call."""
import sys, time, tempfile
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd import ValidateNaturalInference as V
from naturaldiffusion_amd.dit import DiTEngine, flatten_state_dict, XL2
from naturaldiffusion_amd.synth import synthetic_dit_state_dict, synthetic_vae_flat
from naturaldiffusion_amd.vae import VAEDecoder
dev = torch.device("cuda:0"); n = 8
V.device = str(dev)
dit = DiTEngine(flatten_state_dict(synthetic_dit_state_dict(XL2["depth"], XL2["hidden"], seed=0), XL2["depth"], XL2["hidden"]), 2 * n, device=dev, **XL2)
vae = VAEDecoder(synthetic_vae_flat(4), max_batch=n, latent_ch=4, latent_res=32, device=dev)
outdir = Path(tempfile.mkdtemp(prefix="natinf_validate_"))
T = {}
def tick(name, t0):
    torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0
class D:
    max_batch = 2 * n
    def forward(self, z, t, y): return dit(z, t, y)
def decode(latents, path):
    t0 = time.perf_counter(); img = vae(latents); tick("vae", t0)
    t0 = time.perf_counter(); V.save_image_grid(img, outdir / Path(path).name); tick("png", t0)
    return img
V.denoiser_factory, V.decoder_factory = (lambda: D()), (lambda: decode)
for rep in range(4):
    T.clear(); torch.cuda.synchronize(); t0 = time.perf_counter()
    V.natural_inference("ddim", 24)
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    print(f"call {rep}: {tot * 1e3:.1f} ms; vae {T['vae'] * 1e3:.1f}, png {T['png'] * 1e3:.1f}, the rest (set-up + 24 steps) {(tot - T['vae'] - T['png']) * 1e3:.1f}", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); V.natural_inference("ddim", 24); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
