"""Oracle: DiT forward (class-conditional, adaLN-Zero, learn_sigma), torch CPU fp32.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Functional restatement of ``deps/DiT/models.py``:
``modulate`` :19-20, ``TimestepEmbedder`` :27-64, ``LabelEmbedder`` :67-99 (eval: plain table lookup), ``DiTBlock``
:105-126, ``FinalLayer`` :129-146, ``DiT.forward`` / ``unpatchify`` :222-253, fixed 2-D sin-cos position embedding
:279-326.  The three building blocks the reference imports from ``timm`` (``PatchEmbed``, ``Attention``, ``Mlp``;
models.py:16, un-vendored and unpinned) are restated from their published definitions (strided conv patch
embedding; qkv-linear multi-head softmax attention with scale head_dim**-0.5; fc1-GELU(tanh)-fc2) -- that part
of the arithmetic is "parity unpinned"; everything DiT-specific is pinned by ``tests/golden/dit_forward.npz``,
captured from the reference's own ``DiT`` class.

Parameters: flat ``{name: tensor}`` dict with the reference module's state-dict keys."""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


def pos_embed_2d(dim: int, grid: int) -> torch.Tensor:
    """models.py:279-326: [grid*grid, dim] fp32; first half encodes the w index, second half the h index
    (the reference's ``np.meshgrid(grid_w, grid_h)`` puts w first); each half = [sin | cos]."""
    def one(d, pos):
        omega = 1.0 / 10000 ** (np.arange(d // 2, dtype=np.float64) / (d / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    gw, gh = np.meshgrid(np.arange(grid, dtype=np.float32), np.arange(grid, dtype=np.float32))
    emb = np.concatenate([one(dim // 2, gw), one(dim // 2, gh)], axis=1)
    return torch.from_numpy(emb).float()


def param_shapes(depth: int, D: int, in_ch: int = 4, patch: int = 2, grid: int = 16, num_classes: int = 1000,
                 learn_sigma: bool = True) -> Dict[str, tuple]:
    out_ch = in_ch * 2 if learn_sigma else in_ch
    s: Dict[str, tuple] = {"pos_embed": (1, grid * grid, D),
                           "x_embedder.proj.weight": (D, in_ch, patch, patch), "x_embedder.proj.bias": (D,),
                           "t_embedder.mlp.0.weight": (D, 256), "t_embedder.mlp.0.bias": (D,),
                           "t_embedder.mlp.2.weight": (D, D), "t_embedder.mlp.2.bias": (D,),
                           "y_embedder.embedding_table.weight": (num_classes + 1, D)}
    for i in range(depth):
        p = f"blocks.{i}."
        s[p + "attn.qkv.weight"] = (3 * D, D); s[p + "attn.qkv.bias"] = (3 * D,)
        s[p + "attn.proj.weight"] = (D, D); s[p + "attn.proj.bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (4 * D, D); s[p + "mlp.fc1.bias"] = (4 * D,)
        s[p + "mlp.fc2.weight"] = (D, 4 * D); s[p + "mlp.fc2.bias"] = (D,)
        s[p + "adaLN_modulation.1.weight"] = (6 * D, D); s[p + "adaLN_modulation.1.bias"] = (6 * D,)
    s["final_layer.linear.weight"] = (patch * patch * out_ch, D); s["final_layer.linear.bias"] = (patch * patch * out_ch,)
    s["final_layer.adaLN_modulation.1.weight"] = (2 * D, D); s["final_layer.adaLN_modulation.1.bias"] = (2 * D,)
    return s


def make_params(depth: int, D: int, seed: int = 0, **kw) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic weights: xavier-uniform matrices (adaLN and the output layer included, so that the
    modulation path the reference zero-initialises is exercised), N(0, 0.02) embeddings / biases, sin-cos pos_embed."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shp in param_shapes(depth, D, **kw).items():
        if name == "pos_embed":
            out[name] = pos_embed_2d(D, int(round(math.sqrt(shp[1])))).unsqueeze(0)
        elif len(shp) >= 2 and "embedding_table" not in name:
            fan_out, fan_in = shp[0], int(np.prod(shp[1:]))
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            out[name] = (torch.rand(shp, generator=g) * 2 - 1) * lim
        else:
            out[name] = torch.randn(shp, generator=g) * 0.02
    return out


def timestep_embedding(t: torch.Tensor, dim: int = 256, max_period: int = 10000) -> torch.Tensor:
    """models.py:41-58: [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def modulate(x, shift, scale):
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def attention(x, P, pre, heads):
    B, T, D = x.shape
    hd = D // heads
    qkv = F.linear(x, P[pre + "qkv.weight"], P[pre + "qkv.bias"]).reshape(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    w = torch.softmax((q @ k.transpose(-2, -1)) * hd ** -0.5, dim=-1)
    o = (w @ v).transpose(1, 2).reshape(B, T, D)
    return F.linear(o, P[pre + "proj.weight"], P[pre + "proj.bias"])


@torch.no_grad()
def forward(P: Dict[str, torch.Tensor], x: torch.Tensor, t: torch.Tensor, y: torch.Tensor, heads: int, taps=None):
    """DiT.forward (models.py:237-253) in eval mode -> [B, out_ch, H, W]."""
    depth = 1 + max(int(k.split(".")[1]) for k in P if k.startswith("blocks."))
    D = P["pos_embed"].shape[-1]
    w = P["x_embedder.proj.weight"]
    p = w.shape[-1]
    h = F.conv2d(x, w, P["x_embedder.proj.bias"], stride=p).flatten(2).transpose(1, 2) + P["pos_embed"]
    te = F.linear(F.silu(F.linear(timestep_embedding(t), P["t_embedder.mlp.0.weight"], P["t_embedder.mlp.0.bias"])),
                  P["t_embedder.mlp.2.weight"], P["t_embedder.mlp.2.bias"])
    c = te + P["y_embedder.embedding_table.weight"][y.long()]
    if taps is not None:
        taps["embed"] = h; taps["c"] = c
    for i in range(depth):
        pre = f"blocks.{i}."
        m = F.linear(F.silu(c), P[pre + "adaLN_modulation.1.weight"], P[pre + "adaLN_modulation.1.bias"]).chunk(6, dim=1)
        h = h + m[2].unsqueeze(1) * attention(modulate(F.layer_norm(h, (D,), eps=1e-6), m[0], m[1]), P, pre + "attn.", heads)
        z = modulate(F.layer_norm(h, (D,), eps=1e-6), m[3], m[4])
        z = F.linear(F.gelu(F.linear(z, P[pre + "mlp.fc1.weight"], P[pre + "mlp.fc1.bias"]), approximate="tanh"),
                     P[pre + "mlp.fc2.weight"], P[pre + "mlp.fc2.bias"])
        h = h + m[5].unsqueeze(1) * z
        if taps is not None:
            taps[f"block{i}"] = h
    m = F.linear(F.silu(c), P["final_layer.adaLN_modulation.1.weight"], P["final_layer.adaLN_modulation.1.bias"]).chunk(2, dim=1)
    h = F.linear(modulate(F.layer_norm(h, (D,), eps=1e-6), m[0], m[1]), P["final_layer.linear.weight"], P["final_layer.linear.bias"])
    g = int(round(math.sqrt(h.shape[1])))
    oc = h.shape[2] // (p * p)
    h = h.reshape(h.shape[0], g, g, p, p, oc)
    return torch.einsum("nhwpqc->nchpwq", h).reshape(h.shape[0], oc, g * p, g * p)
