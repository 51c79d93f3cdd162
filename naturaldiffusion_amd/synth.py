"""Synthetic NCSN++ weights for benchmarking / smoke runs (the reference ships no checkpoint:
``checkpoint_8.pth`` is a missing large blob).  Recipe: one CPU generator seeded with
``seed``; in engine parameter order, matrices / filters ~ fan-avg uniform, norm scales 1, biases 0, each
then perturbed by ``perturb * randn`` so that the reference's zero-initialised layers are exercised.
Host-side data generation only; returns the flat fp32 vector ``NCSNppEngine`` consumes."""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

from .ncsnpp import param_layout


def synthetic_state_dict(seed: int = 0, perturb: float = 0.01) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shp in param_layout():
        if len(shp) >= 2:
            rf = int(np.prod(shp[2:])) if len(shp) > 2 else 1
            fan_in, fan_out = (shp[0], shp[1]) if ".NIN_" in name else (shp[1] * rf, shp[0] * rf)
            w = (torch.rand(shp, generator=g) * 2 - 1) * math.sqrt(3.0 / ((fan_in + fan_out) / 2))
        elif name.endswith(".weight"):
            w = torch.ones(shp)
        else:
            w = torch.zeros(shp)
        out[name] = (w + perturb * torch.randn(shp, generator=g)).contiguous()
    return out


def synthetic_flat_params(seed: int = 0, perturb: float = 0.01) -> torch.Tensor:
    return torch.cat([t.reshape(-1) for t in synthetic_state_dict(seed, perturb).values()])


def dit_pos_embed(dim: int, grid: int = 16) -> torch.Tensor:
    """Fixed 2-D sin-cos position table [grid*grid, dim] (deps/DiT/models.py:279-326): first half encodes the w index,
    second half the h index, each half = [sin | cos] over dim/4 frequencies 10000^(-i/(dim/4))."""
    def one(d, pos):
        omega = 1.0 / 10000 ** (np.arange(d // 2, dtype=np.float64) / (d / 2.0))
        a = pos.reshape(-1)[:, None] * omega[None]
        return np.concatenate([np.sin(a), np.cos(a)], axis=1)
    gw, gh = np.meshgrid(np.arange(grid, dtype=np.float32), np.arange(grid, dtype=np.float32))
    return torch.from_numpy(np.concatenate([one(dim // 2, gw), one(dim // 2, gh)], axis=1)).float()


def synthetic_dit_state_dict(depth: int = 28, hidden: int = 1152, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Synthetic DiT weights (``DiT-XL-2-256x256.pt`` is a download the image lacks): xavier-uniform matrices --
    including the adaLN / output layers the reference zero-initialises --, N(0, 0.02) embeddings and biases."""
    from .dit import param_layout as dit_layout
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shp in dit_layout(depth, hidden):
        if name == "pos_embed":
            out[name] = dit_pos_embed(hidden).unsqueeze(0)
        elif len(shp) >= 2 and "embedding_table" not in name:
            lim = math.sqrt(6.0 / (int(np.prod(shp[1:])) + shp[0]))
            out[name] = (torch.rand(shp, generator=g) * 2 - 1) * lim
        else:
            out[name] = torch.randn(shp, generator=g) * 0.02
    return out


def synthetic_mmdit_flat(grid: int, layers: int, heads: int, joint_dim: int, pooled_dim: int, in_ch: int = 16, seed: int = 0,
                         pos_max: int = 192, pos_base: int = 64) -> torch.Tensor:
    """Synthetic SD3 MMDiT weights as the flat vector ``MMDiTEngine`` consumes (the SD3-medium checkpoint is a gated
    download): xavier-uniform matrices, N(0, 0.02) biases, the cropped sin-cos position table."""
    from .mmdit import param_layout as mm_layout
    g = torch.Generator().manual_seed(seed)
    parts = []
    D = heads * 64
    for name, shp in mm_layout(layers, heads, joint_dim, pooled_dim, in_ch):
        if name == "pos_embed.pos_embed":
            def one(d, pos):
                omega = 1.0 / 10000 ** (np.arange(d // 2, dtype=np.float64) / (d / 2.0))
                a = pos.reshape(-1)[:, None] * omega[None]
                return np.concatenate([np.sin(a), np.cos(a)], axis=1)
            top = (pos_max - grid) // 2
            ax = (np.arange(pos_max, dtype=np.float32) / (pos_max / pos_base))[top:top + grid]
            gw, gh = np.meshgrid(ax, ax)
            parts.append(torch.from_numpy(np.concatenate([one(D // 2, gw), one(D // 2, gh)], axis=1)).float().reshape(-1))
        elif len(shp) >= 2:
            lim = math.sqrt(6.0 / (int(np.prod(shp[1:])) + shp[0]))
            parts.append(((torch.rand(shp, generator=g) * 2 - 1) * lim).reshape(-1))
        else:
            parts.append((torch.randn(shp, generator=g) * 0.02).reshape(-1))
    return torch.cat(parts)


def synthetic_inception_flat(seed: int = 0) -> torch.Tensor:
    """Stand-in FID Inception-V3 weights (``pt_inception-2015-12-05-6726825d.pth`` is a download the image does not hold): in
    ``inception.param_layout()`` order, variance-preserving uniform filters and non-identity BatchNorm statistics, so that the pool3
    features of random images are O(1) and distinct per image.  Lets the FID epilogue of a 50k-image job RUN (timing, sharding, the
    statistics all-reduce); the number it produces is not a FID -- the callers print ``fid: blocked``."""
    from .inception import param_layout as inc_layout
    g = torch.Generator().manual_seed(1000 + seed)
    parts = []
    for name, shp in inc_layout():
        if name.endswith("conv.weight"):
            fan_in = shp[1] * shp[2] * shp[3]
            t = (torch.rand(shp, generator=g) * 2 - 1) * math.sqrt(6.0 / fan_in)
        elif name.endswith("bn.weight"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif name.endswith("running_var"):
            t = 0.5 + torch.rand(shp, generator=g)
        else:
            t = 0.1 * torch.randn(shp, generator=g)
        parts.append(t.reshape(-1))
    return torch.cat(parts)


def synthetic_vae_flat(latent_ch: int = 4, seed: int = 0) -> torch.Tensor:
    """Stand-in AutoencoderKL decoder weights in ``vae.param_layout`` order behind an identity ``post_quant_conv``: norm scales 1,
    biases small, filters / matrices ~ N(0, 1 / fan_in)."""
    from .vae import param_layout as vae_layout
    g = torch.Generator().manual_seed(2000 + seed)
    parts = [torch.eye(latent_ch).reshape(-1), torch.zeros(latent_ch)]
    for name, shape in vae_layout(latent_ch):
        if name.endswith(("norm1.weight", "norm2.weight", "norm.weight", "norm_out.weight")):
            parts.append(torch.ones(shape).reshape(-1))
        elif len(shape) == 1:
            parts.append(0.01 * torch.randn(shape, generator=g))
        else:
            fan = int(np.prod(shape[1:]))
            parts.append((torch.randn(shape, generator=g) / fan ** 0.5).reshape(-1))
    return torch.cat(parts)
