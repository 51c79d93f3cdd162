"""Drop-in for ``src/SD3NaturalInference.py``: SD3-medium, 28 steps, CFG 7, fp16, seed 10, n = 4.

Public names as in the reference: ``weighted_sum(seq_xstarts, weights=None)``, ``euler_weighted_sum``,
``sd_natural_inference_tx()``, ``sd_euler_natural_inference_tx()``.  Per step the two MMDiT calls are
followed by ONE fused launch (``natinf_step_f16chain``: x0 from velocity, CFG fuse, append, row-normalised
fp16 weighted mean, next model input).  The text encoders / scheduler / VAE are third-party (``diffusers``,
un-vendored and unpinned in the reference); a ``pipe`` object with the same attributes can be passed in.  The
denoiser ``pipe.transformer`` is replaced by the gfx950 MMDiT engine (``use_native_transformer``; on by default when
the pipe is loaded here): text and null prompts then go through ONE batched forward per step.
The reference's 28x5x4 intermediate VAE decodes (:223-236, visualisation only) are off by default.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ptr, stream_ptr
from .coeff import load_sd3_csv, SparseRows
from .sampler import SD3NI

root_path = Path(__file__).resolve().parent.parent
results_path = root_path            # where results/sd3/*.png go (the reference writes below its own root, :241)
PROMPT = "A cat holding a sign that says hello world"


def weighted_sum(seq_xstarts: Sequence[torch.Tensor], weights=None) -> torch.Tensor:
    """Reference :157-168: fp16 chain, row ``len(seq)-1`` of ``weights`` (uniform mean if None)."""
    _lib.require_gpu()
    n = len(seq_xstarts)
    slab = torch.stack([s.contiguous().reshape(-1) for s in seq_xstarts]).to(torch.float16)
    E = slab.shape[1]
    row = np.ones((1, n)) if weights is None else np.asarray(weights, np.float64)[n - 1:n, :n]
    rows = SparseRows(row, lambda k: n, torch.float32, slab.device, dense=True, diag=False)
    out = torch.empty(E, dtype=torch.float16, device=slab.device)
    idx, val, nt = rows.ptrs(0)
    check(lib.natinf_weighted_mean_f16(ptr(slab), ptr(out), idx, val, nt, rows.rows[0].total, E, stream_ptr()),
          "natinf_weighted_mean_f16")
    return out.view(seq_xstarts[0].shape)


def euler_weighted_sum(seq_xstarts, cliplen=0):
    """Reference :61-69: ``seq_xstarts`` = [(weight 0-d tensor, x0)], returns (acc, acc/sum_w) in fp16 ops."""
    _lib.require_gpu()
    seq = seq_xstarts[-cliplen:]
    n = len(seq)
    slab = torch.stack([x.contiguous().reshape(-1) for _, x in seq]).to(torch.float16)
    E = slab.shape[1]
    h = lambda t: float(torch.as_tensor(t, dtype=torch.float32).to(torch.float16))
    tot = 0
    for w, _ in seq:
        tot = tot + torch.as_tensor(w, dtype=torch.float32).cpu()
    row = np.array([[h(torch.as_tensor(w).cpu()) for w, _ in seq]])
    rows = SparseRows(row, lambda k: n, torch.float32, slab.device, dense=True, diag=False)
    idx, val, nt = rows.ptrs(0)
    acc = torch.empty(E, dtype=torch.float16, device=slab.device)
    mean = torch.empty(E, dtype=torch.float16, device=slab.device)
    check(lib.natinf_weighted_mean_f16(ptr(slab), ptr(acc), idx, val, nt, 1.0, E, stream_ptr()), "natinf_weighted_mean_f16")
    # `acc / acc_weight`: a 0-d fp32 divisor keeps its fp32 value (it is only cast to fp16 as a FIRST operand)
    check(lib.natinf_weighted_mean_f16(ptr(slab), ptr(mean), idx, val, nt, float(tot), E, stream_ptr()), "natinf_weighted_mean_f16")
    shape = seq[0][1].shape
    return acc.view(shape), mean.view(shape)


def use_native_transformer(pipe, n: int, latent_side: int = 128, ctx_tokens: Optional[int] = None, device="cuda:0"):
    """Swap ``pipe.transformer`` (diffusers ``SD3Transformer2DModel``) for the HIP engine built from its own weights
    (include/natinf_mmdit.h).  ``n`` = images per batch; the engine is sized for the 2n sequences of a CFG step."""
    from .mmdit import MMDiTEngine, flatten_state_dict
    tr = pipe.transformer
    if isinstance(tr, MMDiTEngine):
        return pipe
    cfg = tr.config
    kw = dict(layers=cfg.num_layers, heads=cfg.num_attention_heads, joint_dim=cfg.joint_attention_dim,
              pooled_dim=cfg.pooled_projection_dim, in_ch=cfg.in_channels)
    if cfg.attention_head_dim != 64 or cfg.patch_size != 2:
        raise ValueError("the engine supports head_dim 64 / patch 2 (SD3-medium)")
    grid = latent_side // 2
    if ctx_tokens is None:
        ctx_tokens = 77 + 256                               # CLIP (77) + T5 (max_sequence_length 256): encode_prompt's default
    flat = flatten_state_dict(tr.state_dict(), grid, **kw)
    pipe.transformer = MMDiTEngine(flat, max_batch=2 * n, grid=grid, ctx_tokens=ctx_tokens, device=device, **kw)
    return pipe


def use_native_vae(pipe, n: int, latent_side: int = 128, device="cuda:0"):
    """Decode with the HIP AutoencoderKL decoder engine (include/natinf_vae.h) built from ``pipe.vae``'s own weights instead of
    ``pipe.vae.decode`` (reference :238-240); ``pipe.vae`` stays in place for its config (scaling / shift factors)."""
    from .vae import VAEDecoder, flatten_state_dict
    if getattr(pipe, "natinf_vae", None) is None:
        lc = int(pipe.vae.config.latent_channels)
        pipe.natinf_vae = VAEDecoder(flatten_state_dict(pipe.vae.state_dict(), lc, prefix="decoder."), max_batch=n,
                                     latent_ch=lc, latent_res=latent_side, device=device)
    return pipe


def _load_pipe(pipe, device, dtype, n=4):
    if pipe is not None:
        return pipe
    try:
        from diffusers import StableDiffusion3Pipeline
    except ImportError as e:
        raise ImportError("SD3 needs the `diffusers` package and the stabilityai/stable-diffusion-3-medium-diffusers "
                          "weights (un-vendored in the reference); pass pipe=... to use another denoiser") from e
    pipe = StableDiffusion3Pipeline.from_pretrained("stabilityai/stable-diffusion-3-medium-diffusers",
                                                    torch_dtype=dtype, local_files_only=True).to(device)
    return use_native_vae(use_native_transformer(pipe, n, device=device), n, device=device)


def philox_noise_f16(indices, shape_per_image=(16, 128, 128), seed: int = 10, device="cuda:0") -> torch.Tensor:
    """fp16 N(0,1) latents, image i keyed by its GLOBAL index (include/natinf.h natinf_randn_philox_f32, rounded to fp16 once): the
    same image for any GPU count / batch split -- the sharded counterpart of the reference's one ``torch.randn(n, 16, 128, 128)`` (:184-186),
    as ``CIFAR10NaturalInference.philox_noise`` is for config 3."""
    from .CIFAR10NaturalInference import philox_noise
    return philox_noise(indices, shape_per_image, seed, device).to(torch.float16)


def _prepare(pipe, device, dtype, n, seed, num_step, noises):
    prompts = [PROMPT] * n
    if noises is None:
        generator = torch.Generator(device)
        generator.manual_seed(seed)
        noises = torch.randn(n, 16, 128, 128, device=device, dtype=dtype, generator=generator)
    emb = pipe.encode_prompt(prompt=prompts, prompt_2=None, prompt_3=None, negative_prompt="")
    pipe.scheduler.set_timesteps(num_step, device=device)
    return noises, emb, pipe.scheduler.timesteps.to(device), pipe.scheduler.sigmas.to(device)


def _decode(pipe, latents):
    """Reference :238-240 -> list of uint8 HWC RGB arrays (the pixels ``image_processor.postprocess(output_type="pil")`` holds:
    ``(x / 2 + 0.5).clamp(0, 1) * 255`` rounded)."""
    z = (latents / pipe.vae.config.scaling_factor) + pipe.vae.config.shift_factor
    dec = getattr(pipe, "natinf_vae", None)
    if dec is not None:
        images = dec(z.float())
        return list(((images * 0.5 + 0.5).clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy())
    images = pipe.vae.decode(z, return_dict=False)[0]
    return [np.array(im) for im in pipe.image_processor.postprocess(images, output_type="pil")]


def _write_row(path, images):
    """Reference :241-243 (``cv2.imwrite`` of the BGR-flipped row of images) without cv2: the same RGB file through PIL."""
    from PIL import Image
    os.makedirs(os.path.dirname(str(path)), exist_ok=True)
    Image.fromarray(np.hstack(images)).save(str(path))


def _velocities(pipe, x, ts, emb):
    pe, ne, ppe, npe = emb
    from .mmdit import MMDiTEngine
    if isinstance(pipe.transformer, MMDiTEngine) and pipe.transformer.max_batch >= 2 * x.shape[0]:
        n = x.shape[0]                                      # text and null prompts of a step in one batched forward
        v = pipe.transformer.forward(torch.cat([x, x]), torch.cat([ts, ts]), torch.cat([pe, ne]), torch.cat([ppe, npe]))
        return v[:n].contiguous(), v[n:].contiguous()
    vt = pipe.transformer(hidden_states=x, timestep=ts, encoder_hidden_states=pe, pooled_projections=ppe, return_dict=False)[0]
    vn = pipe.transformer(hidden_states=x, timestep=ts, encoder_hidden_states=ne, pooled_projections=npe, return_dict=False)[0]
    return vt.contiguous(), vn.contiguous()


@torch.no_grad()
def sd_generate_sharded(pipe, sample_count: int, n: int = 4, rank: int = 0, world: int = 1, seed: int = 10, num_step: int = 28,
                        weight_name: str = "sd3_step_28_weight.csv", device="cuda:0", latent_shape=(16, 128, 128)):
    """Batch-sharded SD3 generation (BASELINE configs 4 / 5): this rank runs the reference's loop (:198-223) for the images
    whose GLOBAL index is rank, rank + world, ... in batches of ``n`` -- no collective on the data path, Philox noise keyed by the global index
    (``philox_noise_f16``), so image i is the same bytes whatever the GPU count or batch split.  ``pipe`` as in ``sd_natural_inference_tx``
    (``encode_prompt`` is asked for ``n`` prompts once; a ragged last batch uses the first rows).
    Returns (final latents [n_local, *latent_shape] fp16 on the device, their global indices [n_local] int64 on the CPU)."""
    from .shard import rank_batches
    _lib.require_gpu()
    dev = torch.device(device)
    batches = list(rank_batches(sample_count, n, rank, world))
    if not batches:
        return torch.empty((0,) + tuple(latent_shape), dtype=torch.float16, device=dev), torch.empty(0, dtype=torch.int64)
    emb = pipe.encode_prompt(prompt=[PROMPT] * n, prompt_2=None, prompt_3=None, negative_prompt="")
    pipe.scheduler.set_timesteps(num_step, device=dev)
    timesteps, sigmas = pipe.scheduler.timesteps.to(dev), pipe.scheduler.sigmas.to(dev)
    weights = load_sd3_csv(os.path.join(root_path / "weights", weight_name))
    samplers, outs = {}, []
    for batch in batches:
        nb = len(batch)
        noises = philox_noise_f16(batch, latent_shape, seed, dev)
        if nb not in samplers:                                          # (a ragged last batch gets its own history slabs)
            samplers[nb] = SD3NI(weights, sigmas, noises.numel(), device=dev, cfg=7.0)
        ni, e = samplers[nb], tuple(t[:nb] for t in emb)
        flat_noise = noises.reshape(-1)
        x = ni.first_input(flat_noise)
        for kk in range(num_step):
            vt, vn = _velocities(pipe, x.view(noises.shape), timesteps[kk].expand(nb), e)
            mean, x = ni.step(kk, x, vt.reshape(-1), vn.reshape(-1), flat_noise, want_next=kk + 1 < num_step)
        outs.append(mean.view(noises.shape).clone())
    return torch.cat(outs), torch.cat([torch.tensor(b, dtype=torch.int64) for b in batches])


@torch.no_grad()
def sd_natural_inference_tx(pipe=None, device="cuda", noises: Optional[torch.Tensor] = None, n: int = 4, seed: int = 10,
                            num_step: int = 28, weight_names=("sd3_step_28_weight.csv", "sd3_step_28_weight_sharp.csv"),
                            decode: bool = True, rank: int = 0, world: int = 1, sample_count: Optional[int] = None, latent_shape=None):
    """Reference :172-245.  Returns the final latents per weight file (and writes ``results/sd3/sgl_*.png``
    when ``decode``).  With ``sample_count`` (not in the reference: its job is one batch of four) the job is ``sample_count`` images sharded by
    global index over ``world`` ranks (``sd_generate_sharded``): returns [(latents of this rank, global indices)] per weight file and, when
    ``decode``, writes this rank's images to ``results/sd3/sgl_<weights>_<index>.png``.  ``latent_shape`` of the sharded job: the native
    transformer's own (in_ch, 2 grid, 2 grid) when ``pipe.transformer`` is the HIP engine, the reference's (16, 128, 128) otherwise."""
    dtype = torch.float16
    pipe = _load_pipe(pipe, device, dtype, n)
    if sample_count is not None:
        if latent_shape is None:
            tr = getattr(pipe, "transformer", None)
            latent_shape = (tr.in_ch, 2 * tr.grid, 2 * tr.grid) if hasattr(tr, "in_ch") and hasattr(tr, "grid") else (16, 128, 128)
        finals = []
        # an index-less "cuda" (this function's default) is THIS rank's current device -- where `_load_pipe` and the engine live after the launcher's
        # `torch.cuda.set_device(local_rank)` -- never a literal cuda:0: rank r's Philox noise and history slabs belong next to rank r's transformer
        dev_r = torch.device(device)
        if dev_r.type == "cuda" and dev_r.index is None:
            dev_r = torch.device("cuda", torch.cuda.current_device())
        for weight_name in weight_names:
            lat, idx = sd_generate_sharded(pipe, sample_count, n, rank, world, seed, num_step, weight_name, dev_r, latent_shape=tuple(latent_shape))
            finals.append((lat, idx))
            if decode:
                for s0 in range(0, lat.shape[0], n):
                    for im, gi in zip(_decode(pipe, lat[s0:s0 + n]), idx[s0:s0 + n].tolist()):
                        _write_row(results_path / ("results/sd3/sgl_%s_%06d.png" % (weight_name[:-4], gi)), [im])
        return finals
    if world != 1 or rank != 0:
        raise ValueError("rank / world need sample_count (the reference's job is one batch: nothing to shard)")
    noises, emb, timesteps, sigmas = _prepare(pipe, device, dtype, n, seed, num_step, noises)
    shape, finals = noises.shape, []
    for weight_name in weight_names:
        weights = load_sd3_csv(os.path.join(root_path / "weights", weight_name))
        ni = SD3NI(weights, sigmas, noises.numel(), device=noises.device, cfg=7.0)
        flat_noise = noises.contiguous().reshape(-1)
        x = ni.first_input(flat_noise)
        for kk in range(num_step):
            ts = timesteps[kk].expand(shape[0])
            vt, vn = _velocities(pipe, x.view(shape), ts, emb)
            mean, x = ni.step(kk, x, vt.reshape(-1), vn.reshape(-1), flat_noise, want_next=kk + 1 < num_step)
        out = mean.view(shape).clone()
        finals.append(out)
        if decode:
            _write_row(results_path / ("results/sd3/sgl_%s.png" % (weight_name[:-4])), _decode(pipe, out))
    return finals


@torch.no_grad()
def sd_euler_natural_inference_tx(pipe=None, device="cuda", noises: Optional[torch.Tensor] = None, n: int = 4,
                                  seed: int = 10, num_step: int = 28, decode: bool = True) -> torch.Tensor:
    """Reference :81-154 with ``is_vanilla_update = False``: flow-Euler written as Natural Inference."""
    dtype = torch.float16
    pipe = _load_pipe(pipe, device, dtype, n)
    noises, emb, timesteps, sigmas = _prepare(pipe, device, dtype, n, seed, num_step, noises)
    shape = noises.shape
    ni = SD3NI(None, sigmas, noises.numel(), device=noises.device, cfg=7.0, euler=True)
    flat_noise = noises.contiguous().reshape(-1)
    x = ni.first_input(flat_noise)
    for i in range(num_step):
        ts = timesteps[i].expand(shape[0])
        vt, vn = _velocities(pipe, x.view(shape), ts, emb)
        mean, x = ni.step(i, x, vt.reshape(-1), vn.reshape(-1), flat_noise, want_next=i + 1 < num_step)
    out = mean.view(shape).clone()
    if decode:
        _write_row(results_path / "results/sd3/euler_sgl_clip0.png", _decode(pipe, out))
    return out


if __name__ == "__main__":
    sd_natural_inference_tx()
