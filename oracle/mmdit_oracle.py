"""Oracle: SD3 MMDiT forward (``pipe.transformer`` of src/SD3NaturalInference.py:111-114,210-213), torch CPU fp32.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

**PARITY UNPINNED.**  The denoiser the reference calls is ``diffusers.SD3Transformer2DModel`` of
``stabilityai/stable-diffusion-3-medium-diffusers`` -- a third-party package the reference neither vendors nor pins
(``requirements.txt:13`` ``diffusers``, no version), absent from this image, with no test, fixture or golden vector in
the reference touching its arithmetic.  This file restates the *published* architecture of
that class (Esser et al. 2024, "Scaling Rectified Flow Transformers", and the diffusers 0.29/0.30 module layout), with
the state-dict key names of the public checkpoint:

  pos_embed        2x2 strided-conv patch embedding + centre crop of a fixed 2-D sin-cos table (pos_embed_max_size^2)
  time_text_embed  sinusoidal(256, [cos|sin]) -> Linear-SiLU-Linear, plus pooled text -> Linear-SiLU-Linear; summed
  context_embedder Linear(joint_attention_dim -> D) on the text tokens
  transformer_blocks.i (JointTransformerBlock): adaLN-Zero on both streams (``norm1``, ``norm1_context``: Linear(SiLU(c))
                   -> shift/scale/gate x2; LayerNorm without affine, eps 1e-6), separate q/k/v projections per stream,
                   ONE softmax attention over the concatenated [image tokens; text tokens] sequence, per-stream output
                   projections, gated residuals, per-stream GELU(tanh) MLPs.  The last block is ``context_pre_only``:
                   its text stream only feeds the attention (``norm1_context`` = AdaLayerNormContinuous: scale, shift).
  norm_out         AdaLayerNormContinuous (scale, shift = chunk(Linear(SiLU(c)), 2)), proj_out, unpatchify.
SD3-medium: 24 blocks, 24 heads x 64, D = 1536, joint dim 4096, pooled dim 2048, 16 latent channels, patch 2, no QK norm.

What anchors it to the reference: the call signature and tensor shapes at the reference's call sites
(hidden_states [B,16,128,128] fp16, timestep [B], encoder_hidden_states [B,333,4096], pooled_projections [B,2048],
returns [B,16,128,128]); everything else is this restatement, and the GPU engine is tested against it.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from .dit_oracle import timestep_embedding


def sincos_table(dim: int, size: int, base: int) -> torch.Tensor:
    """[size*size, dim] 2-D sin-cos table on the grid arange(size) / (size / base) (w index first, each half [sin|cos])."""
    def one(d, pos):
        omega = 1.0 / 10000 ** (np.arange(d // 2, dtype=np.float64) / (d / 2.0))
        a = pos.reshape(-1)[:, None] * omega[None]
        return np.concatenate([np.sin(a), np.cos(a)], axis=1)
    ax = np.arange(size, dtype=np.float32) / (size / base)
    gw, gh = np.meshgrid(ax, ax)
    return torch.from_numpy(np.concatenate([one(dim // 2, gw), one(dim // 2, gh)], axis=1)).float()


def param_shapes(layers: int, heads: int, joint_dim: int, pooled_dim: int, in_ch: int = 16, pos_max: int = 192,
                 head_dim: int = 64) -> Dict[str, tuple]:
    D = heads * head_dim
    s: Dict[str, tuple] = {
        "pos_embed.pos_embed": (1, pos_max * pos_max, D),
        "pos_embed.proj.weight": (D, in_ch, 2, 2), "pos_embed.proj.bias": (D,),
        "time_text_embed.timestep_embedder.linear_1.weight": (D, 256), "time_text_embed.timestep_embedder.linear_1.bias": (D,),
        "time_text_embed.timestep_embedder.linear_2.weight": (D, D), "time_text_embed.timestep_embedder.linear_2.bias": (D,),
        "time_text_embed.text_embedder.linear_1.weight": (D, pooled_dim), "time_text_embed.text_embedder.linear_1.bias": (D,),
        "time_text_embed.text_embedder.linear_2.weight": (D, D), "time_text_embed.text_embedder.linear_2.bias": (D,),
        "context_embedder.weight": (D, joint_dim), "context_embedder.bias": (D,)}
    for i in range(layers):
        p = f"transformer_blocks.{i}."
        last = i == layers - 1
        s[p + "norm1.linear.weight"] = (6 * D, D); s[p + "norm1.linear.bias"] = (6 * D,)
        s[p + "norm1_context.linear.weight"] = ((2 if last else 6) * D, D); s[p + "norm1_context.linear.bias"] = ((2 if last else 6) * D,)
        for n in ("to_q", "to_k", "to_v", "add_k_proj", "add_v_proj", "add_q_proj", "to_out.0"):
            s[p + f"attn.{n}.weight"] = (D, D); s[p + f"attn.{n}.bias"] = (D,)
        if not last:
            s[p + "attn.to_add_out.weight"] = (D, D); s[p + "attn.to_add_out.bias"] = (D,)
        s[p + "ff.net.0.proj.weight"] = (4 * D, D); s[p + "ff.net.0.proj.bias"] = (4 * D,)
        s[p + "ff.net.2.weight"] = (D, 4 * D); s[p + "ff.net.2.bias"] = (D,)
        if not last:
            s[p + "ff_context.net.0.proj.weight"] = (4 * D, D); s[p + "ff_context.net.0.proj.bias"] = (4 * D,)
            s[p + "ff_context.net.2.weight"] = (D, 4 * D); s[p + "ff_context.net.2.bias"] = (D,)
    s["norm_out.linear.weight"] = (2 * D, D); s["norm_out.linear.bias"] = (2 * D,)
    s["proj_out.weight"] = (4 * in_ch, D); s["proj_out.bias"] = (4 * in_ch,)
    return s


def make_params(layers: int, heads: int, joint_dim: int, pooled_dim: int, seed: int = 0, pos_max: int = 192, pos_base: int = 64,
                **kw) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic weights: xavier-uniform matrices, N(0, 0.02) biases, the sin-cos position table."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shp in param_shapes(layers, heads, joint_dim, pooled_dim, pos_max=pos_max, **kw).items():
        if name == "pos_embed.pos_embed":
            out[name] = sincos_table(shp[2], pos_max, pos_base).unsqueeze(0)
        elif len(shp) >= 2:
            lim = math.sqrt(6.0 / (int(np.prod(shp[1:])) + shp[0]))
            out[name] = (torch.rand(shp, generator=g) * 2 - 1) * lim
        else:
            out[name] = torch.randn(shp, generator=g) * 0.02
    return out


def cropped_pos_embed(table: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """centre crop of the [1, S*S, D] table to [1, h*w, D]."""
    S = int(round(math.sqrt(table.shape[1])))
    top, left = (S - h) // 2, (S - w) // 2
    return table.reshape(1, S, S, -1)[:, top:top + h, left:left + w].reshape(1, h * w, -1)


def _lin(P, name, x):
    return F.linear(x, P[name + ".weight"], P[name + ".bias"])


def _ln(x):
    return F.layer_norm(x, (x.shape[-1],), eps=1e-6)


@torch.no_grad()
def forward(P: Dict[str, torch.Tensor], hidden_states: torch.Tensor, timestep: torch.Tensor, encoder_hidden_states: torch.Tensor,
            pooled_projections: torch.Tensor, head_dim: int = 64, taps=None) -> torch.Tensor:
    layers = 1 + max(int(k.split(".")[1]) for k in P if k.startswith("transformer_blocks."))
    D = P["pos_embed.proj.bias"].shape[0]
    H = D // head_dim
    Bn, _, hh, ww = hidden_states.shape
    gh, gw = hh // 2, ww // 2
    x = F.conv2d(hidden_states.float(), P["pos_embed.proj.weight"], P["pos_embed.proj.bias"], stride=2).flatten(2).transpose(1, 2)
    x = x + cropped_pos_embed(P["pos_embed.pos_embed"], gh, gw)
    c = _lin(P, "time_text_embed.timestep_embedder.linear_2", F.silu(_lin(P, "time_text_embed.timestep_embedder.linear_1", timestep_embedding(timestep)))) \
        + _lin(P, "time_text_embed.text_embedder.linear_2", F.silu(_lin(P, "time_text_embed.text_embedder.linear_1", pooled_projections.float())))
    e = _lin(P, "context_embedder", encoder_hidden_states.float())
    sc = F.silu(c)
    Tx = x.shape[1]
    if taps is not None:
        taps["x0"] = x; taps["c"] = c; taps["e0"] = e

    def heads(t):
        return t.reshape(Bn, -1, H, head_dim).transpose(1, 2)

    for i in range(layers):
        p = f"transformer_blocks.{i}."
        last = i == layers - 1
        m = _lin(P, p + "norm1.linear", sc).chunk(6, dim=1)
        nx = _ln(x) * (1 + m[1][:, None]) + m[0][:, None]
        if last:
            cs, cshift = _lin(P, p + "norm1_context.linear", sc).chunk(2, dim=1)
            ne = _ln(e) * (1 + cs[:, None]) + cshift[:, None]
        else:
            n = _lin(P, p + "norm1_context.linear", sc).chunk(6, dim=1)
            ne = _ln(e) * (1 + n[1][:, None]) + n[0][:, None]
        q = torch.cat([_lin(P, p + "attn.to_q", nx), _lin(P, p + "attn.add_q_proj", ne)], dim=1)
        k = torch.cat([_lin(P, p + "attn.to_k", nx), _lin(P, p + "attn.add_k_proj", ne)], dim=1)
        v = torch.cat([_lin(P, p + "attn.to_v", nx), _lin(P, p + "attn.add_v_proj", ne)], dim=1)
        w = torch.softmax((heads(q) @ heads(k).transpose(-2, -1)) * head_dim ** -0.5, dim=-1)
        o = (w @ heads(v)).transpose(1, 2).reshape(Bn, -1, D)
        x = x + m[2][:, None] * _lin(P, p + "attn.to_out.0", o[:, :Tx])
        z = _ln(x) * (1 + m[4][:, None]) + m[3][:, None]
        x = x + m[5][:, None] * _lin(P, p + "ff.net.2", F.gelu(_lin(P, p + "ff.net.0.proj", z), approximate="tanh"))
        if not last:
            e = e + n[2][:, None] * _lin(P, p + "attn.to_add_out", o[:, Tx:])
            z = _ln(e) * (1 + n[4][:, None]) + n[3][:, None]
            e = e + n[5][:, None] * _lin(P, p + "ff_context.net.2", F.gelu(_lin(P, p + "ff_context.net.0.proj", z), approximate="tanh"))
        if taps is not None:
            taps[f"x{i + 1}"] = x
            taps[f"e{i + 1}"] = e
    s, sh = _lin(P, "norm_out.linear", sc).chunk(2, dim=1)
    x = _lin(P, "proj_out", _ln(x) * (1 + s[:, None]) + sh[:, None])
    oc = x.shape[2] // 4
    x = x.reshape(Bn, gh, gw, 2, 2, oc)
    return torch.einsum("nhwpqc->nchpwq", x).reshape(Bn, oc, gh * 2, gw * 2)


def flops_per_sequence(layers: int, heads: int, joint_dim: int, pooled_dim: int, tx: int, tc: int, head_dim: int = 64) -> float:
    """2*MAC of the matmuls of one forward of one sequence."""
    D = heads * head_dim
    T = tx + tc
    f = 2.0 * (tx * D * 64 + D * 256 + D * D + D * pooled_dim + D * D + tc * joint_dim * D)
    for i in range(layers):
        last = i == layers - 1
        f += 2.0 * D * (6 * D + (2 if last else 6) * D)                         # adaLN linears
        f += 2.0 * T * 3 * D * D + 4.0 * T * T * D                               # q,k,v + QK^T + PV
        f += 2.0 * tx * D * D + 2.0 * tx * 8 * D * D                             # to_out, MLP
        if not last:
            f += 2.0 * tc * D * D + 2.0 * tc * 8 * D * D
    return f + 2.0 * D * 2 * D + 2.0 * tx * D * 64
