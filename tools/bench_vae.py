"""Decode time of the AutoencoderKL decoder engine (GPU box).  usage: bench_vae.py [latent_res:latent_ch:B ...]"""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd.vae import VAEDecoder, param_layout

def synth(latent_ch):
    g = torch.Generator().manual_seed(0)
    parts = [torch.eye(latent_ch).reshape(-1), torch.zeros(latent_ch)]
    for name, shape in param_layout(latent_ch):
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("norm.weight") or name.endswith("norm_out.weight"):
            parts.append(torch.ones(shape).reshape(-1))
        elif len(shape) == 1:
            parts.append(0.01 * torch.randn(shape, generator=g))
        else:
            fan = 1
            for d in shape[1:]: fan *= d
            parts.append((torch.randn(shape, generator=g) / fan ** 0.5).reshape(-1))
    return torch.cat(parts)

cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(32, 4, 8), (64, 16, 4), (128, 16, 1), (128, 16, 4)]
for r, ch, B in cases:
    dec = VAEDecoder(synth(ch), max_batch=B, latent_ch=ch, latent_res=r)
    z = torch.randn(B, ch, r, r, device="cuda")
    out = dec(z); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n): out = dec(z)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"latents {r}x{r}x{ch} B={B}: {ms:.1f} ms per decode, {ms / B:.1f} ms per image, finite={bool(torch.isfinite(out).all())}", flush=True)
    del dec
