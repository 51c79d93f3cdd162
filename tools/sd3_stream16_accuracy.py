"""What the IEEE-half image stream (natinf_set_mmdit_stream16) changes at SD3 size: engine against engine on identical inputs -- one MMDiT forward of 8 sequences (velocity) and the
final latents of a 28-step SD3-form Natural Inference run (4 images x CFG 7, weights/sd3_step_28_weight.csv), fp32 stream against half stream, bf16 and fp8 operands; next to it the
figure the same comparison gives for fp8 against bf16 operands (the error the engine already carries).  Synthetic SD3-medium-shaped weights.  (GPU box)"""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from naturaldiffusion_amd.coeff import load_sd3_csv
from naturaldiffusion_amd.mmdit import MMDiTEngine, SD3_MEDIUM
from naturaldiffusion_amd.sampler import SD3NI
from naturaldiffusion_amd.synth import synthetic_mmdit_flat
dev = torch.device("cuda:0")
cfg = dict(SD3_MEDIUM)
flat = synthetic_mmdit_flat(64, seed=0, **cfg)
g = torch.Generator(device=dev).manual_seed(0)
n, nstep = 4, 28
x0 = torch.randn(2 * n, 16, 128, 128, device=dev, generator=g); t0 = torch.rand(2 * n, device=dev, generator=g) * 1000
text = torch.randn(2 * n, 333, cfg["joint_dim"], device=dev, generator=g); pooled = torch.randn(2 * n, cfg["pooled_dim"], device=dev, generator=g)
noises = torch.randn(n, 16, 128, 128, device=dev, generator=g).half()
W = load_sd3_csv(ROOT / "weights" / "sd3_step_28_weight.csv")
u = np.linspace(1.0, 3 * 0.001 / (1 + 2 * 0.001), nstep)
sig = np.append(3 * u / (1 + 2 * u), 0.0).astype(np.float32)
sigmas, timesteps = torch.from_numpy(sig).to(dev), torch.from_numpy(sig[:-1] * 1000).to(dev)


def run(fp8, s16):
    eng = MMDiTEngine(flat, max_batch=2 * n, grid=64, ctx_tokens=333, device=dev, fp8=fp8, stream16=s16, **cfg)
    v = eng.forward(x0, t0, text, pooled).float().clone()
    ni = SD3NI(W, sigmas, noises.numel(), device=dev, cfg=7.0)
    z = noises.reshape(-1)
    x = ni.first_input(z)
    for k in range(nstep):
        xx = x.view(n, 16, 128, 128)
        vv = eng.forward(torch.cat([xx, xx]), timesteps[k].expand(2 * n), text, pooled)
        mean, x = ni.step(k, x, vv[:n].reshape(-1), vv[n:].reshape(-1), z, want_next=k + 1 < nstep)
    lat = mean.float().clone()
    del eng
    return v, lat


rel = lambda a, b: (float(((a - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt()), float((a - b).abs().max() / b.abs().max()))
res = {(fp8, s16): run(fp8, s16) for fp8 in (False, True) for s16 in (False, True)}
for fp8 in (False, True):
    (v32, l32), (v16, l16) = res[(fp8, False)], res[(fp8, True)]
    print(f"{'fp8 ' if fp8 else 'bf16'} operands, half stream against fp32 stream: one forward rel rms {rel(v16, v32)[0]:.3e} (max {rel(v16, v32)[1]:.3e}); "
          f"28-step final latents rel rms {rel(l16, l32)[0]:.3e} (max {rel(l16, l32)[1]:.3e}); finite: {bool(torch.isfinite(l16).all())}")
(vb, lb), (vf, lf) = res[(False, False)], res[(True, False)]
print(f"for scale -- fp8 against bf16 operands (fp32 stream): one forward rel rms {rel(vf, vb)[0]:.3e}; 28-step final latents rel rms {rel(lf, lb)[0]:.3e}")
