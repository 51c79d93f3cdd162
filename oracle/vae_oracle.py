"""Oracle: AutoencoderKL DECODER forward (``vae.decode`` of src/ValidateNaturalInference.py:231-236,298-303,366-371 and
src/SD3NaturalInference.py:139,148,229,239), torch CPU fp32.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

**PARITY UNPINNED.**  The reference takes this module from ``diffusers`` (``AutoencoderKL.from_pretrained`` of
``stabilityai/sd-vae-ft-ema`` / the SD3 pipeline's VAE) -- un-vendored, un-pinned, absent from this image, and no test or
fixture of the reference touches its arithmetic.  This file restates the published architecture of the decoder (Rombach et
al. 2022 first-stage model; diffusers' ``Decoder`` module layout and state-dict key names):

  conv_in 3x3 (latent_ch -> 512); mid_block: ResnetBlock, single-head self-attention over the pixels (GroupNorm ->
  q, k, v Linear -> softmax(q k^T / sqrt(C)) v -> Linear, + residual), ResnetBlock; up_blocks 0..3 with 3 ResnetBlocks each
  (channels 512, 512, 256, 128; a 1x1 ``conv_shortcut`` where the channel count changes) and, after blocks 0..2, nearest 2x
  up-sampling + 3x3 conv; conv_norm_out (GroupNorm) -> SiLU -> conv_out 3x3 (128 -> 3).
  ResnetBlock: x + conv2(SiLU(norm2(conv1(SiLU(norm1(x)))))) (no time embedding in the VAE); GroupNorm: 32 groups, eps 1e-6.
Output: [B, 3, 8r, 8r] for latents [B, latent_ch, r, r].  Scaling (z / scaling_factor + shift_factor) is the caller's, as in
the reference.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

CH = (512, 512, 256, 128)          # up_blocks 0..3 (reversed block_out_channels [128, 256, 512, 512])


def param_shapes(latent_ch: int = 4) -> Dict[str, tuple]:
    s: Dict[str, tuple] = {"conv_in.weight": (512, latent_ch, 3, 3), "conv_in.bias": (512,)}

    def res(p, cin, cout):
        s[p + "norm1.weight"] = (cin,); s[p + "norm1.bias"] = (cin,)
        s[p + "conv1.weight"] = (cout, cin, 3, 3); s[p + "conv1.bias"] = (cout,)
        s[p + "norm2.weight"] = (cout,); s[p + "norm2.bias"] = (cout,)
        s[p + "conv2.weight"] = (cout, cout, 3, 3); s[p + "conv2.bias"] = (cout,)
        if cin != cout:
            s[p + "conv_shortcut.weight"] = (cout, cin, 1, 1); s[p + "conv_shortcut.bias"] = (cout,)
    res("mid_block.resnets.0.", 512, 512)
    a = "mid_block.attentions.0."
    s[a + "group_norm.weight"] = (512,); s[a + "group_norm.bias"] = (512,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        s[a + n + ".weight"] = (512, 512); s[a + n + ".bias"] = (512,)
    res("mid_block.resnets.1.", 512, 512)
    cin = 512
    for i, cout in enumerate(CH):
        for j in range(3):
            res(f"up_blocks.{i}.resnets.{j}.", cin if j == 0 else cout, cout)
        if i < 3:
            s[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (cout, cout, 3, 3); s[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (cout,)
        cin = cout
    s["conv_norm_out.weight"] = (128,); s["conv_norm_out.bias"] = (128,)
    s["conv_out.weight"] = (3, 128, 3, 3); s["conv_out.bias"] = (3,)
    return s


def make_params(latent_ch: int = 4, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic weights: fan-in-scaled uniform filters / matrices, norm scales 1 + 0.1 N(0,1), biases 0.02 N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shp in param_shapes(latent_ch).items():
        if len(shp) >= 2:
            lim = math.sqrt(3.0 / int(np.prod(shp[1:])))
            out[name] = (torch.rand(shp, generator=g) * 2 - 1) * lim
        elif "norm" in name and name.endswith("weight"):
            out[name] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            out[name] = 0.02 * torch.randn(shp, generator=g)
    return out


def _gn(P, p, x):
    return F.group_norm(x, 32, P[p + ".weight"], P[p + ".bias"], eps=1e-6)


def _res(P, p, x):
    h = F.conv2d(F.silu(_gn(P, p + "norm1", x)), P[p + "conv1.weight"], P[p + "conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_gn(P, p + "norm2", h)), P[p + "conv2.weight"], P[p + "conv2.bias"], padding=1)
    if p + "conv_shortcut.weight" in P:
        x = F.conv2d(x, P[p + "conv_shortcut.weight"], P[p + "conv_shortcut.bias"])
    return x + h


def _attn(P, p, x):
    B, C, H, W = x.shape
    h = _gn(P, p + "group_norm", x).reshape(B, C, H * W).transpose(1, 2)            # [B, T, C]
    q = F.linear(h, P[p + "to_q.weight"], P[p + "to_q.bias"])
    k = F.linear(h, P[p + "to_k.weight"], P[p + "to_k.bias"])
    v = F.linear(h, P[p + "to_v.weight"], P[p + "to_v.bias"])
    w = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(C), dim=-1)
    o = F.linear(w @ v, P[p + "to_out.0.weight"], P[p + "to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(B, C, H, W)


@torch.no_grad()
def decode(P: Dict[str, torch.Tensor], z: torch.Tensor, taps=None) -> torch.Tensor:
    z = z.float()
    if "post_quant_conv.weight" in P:                    # AutoencoderKL.decode: z = post_quant_conv(z); dec = decoder(z)
        z = F.conv2d(z, P["post_quant_conv.weight"].reshape(z.shape[1], z.shape[1], 1, 1), P["post_quant_conv.bias"])
    x = F.conv2d(z, P["conv_in.weight"], P["conv_in.bias"], padding=1)
    x = _res(P, "mid_block.resnets.0.", x)
    x = _attn(P, "mid_block.attentions.0.", x)
    x = _res(P, "mid_block.resnets.1.", x)
    if taps is not None:
        taps["mid"] = x
    for i in range(4):
        for j in range(3):
            x = _res(P, f"up_blocks.{i}.resnets.{j}.", x)
        if i < 3:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, P[f"up_blocks.{i}.upsamplers.0.conv.weight"], P[f"up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
        if taps is not None:
            taps[f"up{i}"] = x
    x = F.silu(_gn(P, "conv_norm_out", x))
    return F.conv2d(x, P["conv_out.weight"], P["conv_out.bias"], padding=1)
