"""Drop-in for ``src/ValidateNaturalInference.py``: original DDPM / DDIM skip samplers vs their
Natural Inference (coefficient-matrix) form, DiT-XL/2 + CFG 4.0, 24 steps, seed 0.

Same public names as the reference (``space_timesteps``, ``create_ddpm_coeff``, ``skip_ddpm_coeff``,
``create_ddim_coeff``, ``skip_ddim_coeff``, ``calc_x0_mean_z``, ``forward_cfg``, ``weighted_sum``,
``ddpm_skip_sample``, ``ddim_skip_sample``, ``natural_inference``, ``compare_output_tx``; globals
``vae_path``, ``model_path``).  ``natural_inference`` -- the path being accelerated -- runs one fused
``natinf_step_f32prod`` launch per step.  The two *original* samplers are the baselines it is compared
with and stay host-sequenced tensor algebra.  The DiT-XL/2 denoiser is the gfx950 engine of
``include/natinf_dit.h`` loaded from ``model_path`` (or whatever ``denoiser_factory`` returns); the VAE decoder is
the engine of ``include/natinf_vae.h`` loaded from ``vae_path`` (or ``decoder_factory``), images written with PIL.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Callable, List, Optional

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ptr, stream_ptr
from .coeff import load_coeff_npz, SparseRows
from .sampler import ValidateNI

root_path = Path(__file__).resolve().parent.parent
vae_path = None
model_path = None
device = "cuda:0"

# hooks: () -> object with .forward(z, t, y) returning [B, 8, H, W]; () -> callable(latents) -> images
denoiser_factory: Optional[Callable] = None
decoder_factory: Optional[Callable] = None
last_latents: Optional[torch.Tensor] = None      # final latents of the most recent sampler call


def make_path(path):
    path = os.path.abspath(path)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    return path


def space_timesteps(num_timesteps, section_counts):
    """Timesteps kept when striding a ``num_timesteps`` process (reference :28-79; improved-DDPM recipe)."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(v) for v in section_counts.split(",")]
    base, extra = divmod(num_timesteps, len(section_counts))
    start, steps = 0, []
    for i, count in enumerate(section_counts):
        size = base + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            steps.append(start + round(pos))
            pos += stride
        start += size
    return set(steps)


def _abar():
    betas = np.linspace(0.0001, 0.02, 1000, dtype=np.float64)
    alphas = 1 - betas
    return betas, alphas, np.cumprod(alphas)


def create_ddpm_coeff():
    """[alphas, alphas_bar, log_var, coeff_xt2x0, coeff_eps2x0, coeff_xt, coeff_x0] (reference :82-99)."""
    betas, alphas, abar = _abar()
    prev = np.append(1.0, abar[:-1])
    var = betas * (1.0 - prev) / (1.0 - abar)
    return [alphas, abar, np.log(np.append(1E-5, var[1:])), np.sqrt(1.0 / abar), np.sqrt(1.0 / abar - 1),
            np.sqrt(alphas) * (1 - prev) / (1 - abar), np.sqrt(prev) * betas / (1 - abar)]


def _skip(abar, num_step):
    idx = sorted(space_timesteps(1000, str(num_step)))
    sab = abar[idx]
    sa = np.zeros_like(sab)
    sa[0] = sab[0]
    sa[1:] = sab[1:] / sab[:-1]
    return idx, sa, sab, np.append(1.0, sab[:-1])


def skip_ddpm_coeff(coeff_all, num_step=50):
    """Re-derive the DDPM tables on the strided schedule (reference :102-133)."""
    idx, sa, sab, prev = _skip(coeff_all[1], num_step)
    sb = 1 - sa
    var = sb * (1.0 - prev) / (1.0 - sab)
    out = [sa, sab, np.log(np.append(1E-5, var[1:])), np.sqrt(1.0 / sab), np.sqrt(1.0 / sab - 1),
           np.sqrt(sa) * (1 - prev) / (1 - sab), np.sqrt(prev) * sb / (1 - sab)]
    return out, idx


def create_ddim_coeff():
    """(alphas, alphas_bar, coeff_xt2x0, coeff_eps2x0, coeff_xt, coeff_x0) (reference :136-151)."""
    _, alphas, abar = _abar()
    prev = np.append(1.0, abar[:-1])
    rect = np.sqrt((1 - prev) / (1 - abar))
    return alphas, abar, np.sqrt(1.0 / abar), np.sqrt(1.0 / abar - 1), rect, np.sqrt(prev) - rect * np.sqrt(abar)


def skip_ddim_coeff(coeff_all, num_step=50):
    """Reference :154-174."""
    idx, sa, sab, prev = _skip(coeff_all[1], num_step)
    rect = np.sqrt((1 - prev) / (1 - sab))
    return (sa, sab, np.sqrt(1.0 / sab), np.sqrt(1.0 / sab - 1), rect, np.sqrt(prev) - rect * np.sqrt(sab)), idx


def calc_x0_mean_z(input_z, eps, coeff, ii):
    """Reference :177-182."""
    coeff_xt2x0, coeff_eps2x0, coeff_xt, coeff_x0 = coeff
    x0 = coeff_xt2x0[ii] * input_z - coeff_eps2x0[ii] * eps
    return x0, coeff_xt[ii] * input_z + coeff_x0[ii] * x0


@torch.no_grad()
def _cond_uncond(model, zt, timesteps, classlabels, classnulls):
    """The two denoiser calls of a CFG step (reference :190-191).  A denoiser that takes 2n samples (the gfx950 DiT engine: ``max_batch``) gets them
    as ONE forward of [z; z] with [labels; nulls] -- at n = 8 a DiT-XL/2 forward is ~300 dependent launches of 10-30 us each, so two forwards
    of 8 cost almost twice one of 16; samples are independent inside the engine, so the halves are what the two calls return."""
    n = len(zt)
    if getattr(model, "max_batch", 0) >= 2 * n:
        both = model.forward(torch.cat([zt, zt]), torch.cat([timesteps, timesteps]), torch.cat([classlabels.to(classnulls.dtype), classnulls]))
        return both[:n], both[n:]
    return model.forward(zt, timesteps, classlabels), model.forward(zt, timesteps, classnulls)


@torch.no_grad()
def forward_cfg(model, zt, timesteps, classlabels, cfg_scale, cls):
    """Reference :185-195: two denoiser calls (batched into one where the denoiser allows), first 4 of 8 channels, CFG fuse."""
    classnulls = torch.tensor([cls] * len(zt), device=zt.device)
    cond, uncond = _cond_uncond(model, zt, timesteps, classlabels, classnulls)
    cond_eps, uncond_eps = cond[:, :4, :, :], uncond[:, :4, :, :]
    return cond_eps, uncond_eps, uncond_eps + cfg_scale * (cond_eps - uncond_eps)


@torch.no_grad()
def weighted_sum(weights, seq_elem):
    """Reference :198-204: fp32 products accumulated in fp64 -> fp32 (``natinf_weighted_sum_f32prod``)."""
    _lib.require_gpu()
    slab = torch.stack([s.contiguous().reshape(-1) for s in seq_elem]).to(torch.float32)
    n, E = slab.shape
    rows = SparseRows(np.asarray(weights, np.float64)[None, :n], lambda k: n, torch.float32, slab.device, dense=True, diag=False)
    out = torch.empty(E, dtype=torch.float32, device=slab.device)
    idx, val, nt = rows.ptrs(0)
    check(lib.natinf_weighted_sum_f32prod(ptr(slab), ptr(out), idx, val, nt, E, stream_ptr()), "natinf_weighted_sum_f32prod")
    return out.view(seq_elem[0].shape)


_engine_cache = {}


def load_dit_engine(path, max_batch=16):
    """Reference :150-154 (``DiT_models['DiT-XL/2'](input_size=32, num_classes=1000)`` + ``load_state_dict``) on the
    gfx950 engine: the checkpoint's tensors go straight into ``natinf_dit_load``; one engine per checkpoint path."""
    from .dit import DiTEngine, flatten_state_dict, XL2
    key = (str(path), max_batch)
    if key not in _engine_cache:
        sd = torch.load(path, map_location="cpu", weights_only=True)
        _engine_cache[key] = DiTEngine(flatten_state_dict(sd, XL2["depth"], XL2["hidden"]), max_batch, device=device, **XL2)
    return _engine_cache[key]


def _setup(seed):
    torch.manual_seed(seed)
    torch.set_grad_enabled(False)
    if denoiser_factory is not None:
        model = denoiser_factory()
    elif model_path is not None:
        model = load_dit_engine(model_path)
    else:
        raise RuntimeError("set ValidateNaturalInference.model_path (a DiT-XL/2 state dict, reference :152-154) or "
                           "ValidateNaturalInference.denoiser_factory; see INTEGRATION.md")
    labels = torch.tensor([207, 360, 387, 974, 88, 979, 417, 279], device=device)
    return model, labels, len(labels)


def load_vae_decoder(path, max_batch=8):
    """Reference :212-214 (``AutoencoderKL.from_pretrained(vae_path)``) on the gfx950 decoder engine (include/natinf_vae.h):
    ``path`` is the model directory (``diffusion_pytorch_model.safetensors`` / ``.bin``) or a weights file."""
    from .vae import VAEDecoder, flatten_state_dict
    key = ("vae", str(path), max_batch)
    if key not in _engine_cache:
        p = Path(path)
        if p.is_dir():
            cand = [p / "diffusion_pytorch_model.safetensors", p / "diffusion_pytorch_model.bin"]
            p = next((c for c in cand if c.exists()), cand[0])
        if p.suffix == ".safetensors":
            from safetensors.torch import load_file
            sd = load_file(str(p))
        else:
            sd = torch.load(p, map_location="cpu", weights_only=True)
        _engine_cache[key] = VAEDecoder(flatten_state_dict(sd, 4, prefix="decoder."), max_batch, latent_ch=4, latent_res=32, device=device)
    return _engine_cache[key]


_png_pool = None


def write_png_rgb(arr: np.ndarray, path, level: int = 1, threads: int = 8) -> None:
    """An 8-bit RGB PNG of ``arr`` [H, W, 3] uint8, deflated in ``threads`` row bands at once.  The file is what any PNG reader decodes to the same pixels: filter
    type 0 on every row, ONE zlib stream assembled from independently compressed bands (each band a raw deflate stream closed by a byte-aligned sync flush, the
    last one finished; the zlib header and the Adler-32 of the whole filtered image around them -- the way pigz builds a stream).  Why: the 2066 x 260 row of eight
    256 x 256 images is 1.6 MB of poorly compressible pixels, and PIL's single-threaded encoder took 51-62 ms for it on the GPU box's host -- a quarter of a 24-step
    Validate run of 8 images on the DiT engine, all of it behind the last kernel.  ``zlib.compress`` releases the GIL, so the bands really run side by side."""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    a = np.ascontiguousarray(arr, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("write_png_rgb expects [H, W, 3] uint8")
    h, w = int(a.shape[0]), int(a.shape[1])
    raw = np.zeros((h, 1 + 3 * w), np.uint8)                          # filter byte 0 (None) in front of every row
    raw[:, 1:] = a.reshape(h, 3 * w)
    nb = max(1, min(int(threads), h // 16 if h >= 16 else 1))
    cuts = [h * i // nb for i in range(nb + 1)]
    bands = [raw[cuts[i]:cuts[i + 1]].tobytes() for i in range(nb)]

    def deflate(i):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)                # raw deflate
        return c.compress(bands[i]) + c.flush(zlib.Z_FINISH if i == nb - 1 else zlib.Z_SYNC_FLUSH)
    if nb > 1:
        global _png_pool
        if _png_pool is None:                                         # (kept: starting eight threads costs as much as they save on one image row)
            _png_pool = ThreadPoolExecutor(max(8, nb), thread_name_prefix="natinf-png")
        parts = list(_png_pool.map(deflate, range(nb)))
    else:
        parts = [deflate(0)]
    adler = 1
    for b in bands:
        adler = zlib.adler32(b, adler)
    idat = b"\x78\x01" + b"".join(parts) + struct.pack(">I", adler & 0xffffffff)

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(str(path), "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", idat) + chunk(b"IEND", b""))


def save_image_grid(images: torch.Tensor, path, nrow: int = 8) -> None:
    """``torchvision.utils.save_image(samples, path, nrow=8, normalize=True, value_range=(-1, 1))`` (reference :236; the 8
    validation images come out as one row of 8) without
    torchvision: clamp to [-1, 1], map to [0, 255], tile with 2-pixel padding -- the same pixels -- and write the PNG with ``write_png_rgb`` (lossless, like
    torchvision's PIL writer; deflated at level 1 in parallel row bands instead of level 6 on one thread: 51-62 -> a few ms on the GPU box's host)."""
    # the pixel arithmetic where the tensor lives (on the device for the engines' output: 1.5 MB of uint8 cross to the host instead of 6 MB of fp32, and no
    # OpenMP team of the host's torch spins up behind the last kernel -- under a container's CPU quota that team, not the work, was what stalled a run); the same
    # fp32 operations in the same order as save_image's normalize + make_grid + mul(255).add_(0.5).clamp_(0, 255).to(uint8); pad pixels are 0
    x = images.detach().float()
    u8 = ((((x.clamp(-1, 1) + 1) * 0.5) * 255 + 0.5).clamp(0, 255)).to(torch.uint8).cpu().numpy()
    n, c, h, w = u8.shape
    ncol = min(nrow, n); nr = (n + ncol - 1) // ncol
    arr = np.zeros((nr * (h + 2) + 2, ncol * (w + 2) + 2, 3), np.uint8)
    for i in range(n):
        r0, c0 = (i // ncol) * (h + 2) + 2, (i % ncol) * (w + 2) + 2
        arr[r0:r0 + h, c0:c0 + w, :] = np.transpose(u8[i], (1, 2, 0)) if c == 3 else np.repeat(np.transpose(u8[i], (1, 2, 0)), 3, axis=2)
    write_png_rgb(arr, path, threads=4)


def _finish(input_z, name):
    global last_latents
    last_latents = input_z
    if decoder_factory is not None:
        decoder_factory()(input_z / 0.18215, make_path(root_path / ("results/validation/" + name)))
    elif vae_path is not None:                               # reference :231-236: decode the latents, write the 1x8 image row
        images = load_vae_decoder(vae_path)(input_z / 0.18215)
        save_image_grid(images, make_path(root_path / ("results/validation/" + name)))
    return input_z


def _original(num_step, stochastic, seed=0):
    model, labels, n = _setup(seed)
    tables, skip_idxs = (skip_ddpm_coeff(create_ddpm_coeff(), num_step) if stochastic
                         else skip_ddim_coeff(create_ddim_coeff(), num_step))
    tb = [torch.from_numpy(e).to(device=device, dtype=torch.float32) for e in tables]
    log_var = tb[2] if stochastic else None
    coeff = tuple(tb[3:7]) if stochastic else tuple(tb[2:6])
    input_z = torch.randn(n, 4, 32, 32, device=device)
    for ii in list(range(0, num_step))[::-1]:
        timesteps = torch.ones(n, dtype=torch.int32, device=device) * skip_idxs[ii]
        _, _, fuse_eps = forward_cfg(model, input_z, timesteps, labels, 4.0, 1000)
        _, mean_z = calc_x0_mean_z(input_z, fuse_eps, coeff, ii)
        if stochastic:
            noise = torch.randn_like(input_z, dtype=torch.float32, device=device)
            input_z = mean_z + torch.exp(0.5 * log_var[ii]) * noise
        else:
            input_z = mean_z
    return input_z


def ddpm_skip_sample(num_step=24):
    """Reference :207-256 (the classical ancestral sampler NI is compared with)."""
    return _finish(_original(num_step, True), "ddpm_%03d__seed_%d__original.png" % (num_step, 0))


@torch.no_grad()
def ddim_skip_sample(num_step=24):
    """Reference :259-308."""
    return _finish(_original(num_step, False), "ddim_%03d__seed_%d__original.png" % (num_step, 0))


def natural_inference(alg_name="ddpm", num_step=24):
    """Reference :311-372 on the fused kernel: per step two denoiser calls, one fresh ``randn_like`` written
    straight into the noise-history slab, one ``natinf_step_f32prod`` launch."""
    model, labels, n = _setup(0)
    weight_path = root_path / ("results/%s/%s_%03d.npz" % (alg_name.replace("_sympy", ""), alg_name, num_step))
    C, B, node = load_coeff_npz(weight_path)
    num_step = B.shape[0]
    tables, _ = skip_ddim_coeff(create_ddim_coeff(), num_step)
    c1 = np.asarray(tables[2])[::-1]
    c2 = np.asarray(tables[3])[::-1]
    E = n * 4 * 32 * 32
    ni = ValidateNI(C, B, node, c1.astype(np.float32), c2.astype(np.float32), E, device=device)
    noise = torch.randn(n, 4, 32, 32, device=device)
    ni.hist_eps[0].copy_(noise.reshape(-1))
    input_z = noise.clone()
    classnulls = torch.full((n,), 1000, dtype=labels.dtype, device=device)             # (reference :352 builds it per step from a host list: a blocking copy each time)
    for kk in range(num_step):
        timesteps = torch.full((n,), int(node[kk, 0]), dtype=torch.int32, device=device)
        cond, uncond = _cond_uncond(model, input_z, timesteps, labels, classnulls)      # [n, 8, 32, 32] each; first 4 channels used
        ni.hist_eps[kk + 1].copy_(torch.randn_like(input_z, dtype=torch.float32, device=device).reshape(-1))
        per, stride = 4 * 32 * 32, cond.shape[1] * 32 * 32
        z = ni.step(kk, input_z.reshape(-1), cond.contiguous(), uncond.contiguous(), 4.0, per, stride)
        input_z = z.view(n, 4, 32, 32)
    weight_name = os.path.basename(weight_path)[:-4]
    return _finish(input_z.clone(), "%s__seed_%d__natural.png" % (weight_name, 0))


def compare_output_tx():
    ddpm_skip_sample(24)
    ddim_skip_sample(24)
    natural_inference("ddpm_sympy", 24)
    natural_inference("ddim_sympy", 24)


if __name__ == "__main__":
    vae_path = "./sd-vae-ft-ema"
    model_path = "./DiT-XL-2-256x256.pt"
    compare_output_tx()
