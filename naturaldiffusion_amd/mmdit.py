"""Host wrapper of the gfx950 SD3 MMDiT engine (include/natinf_mmdit.h).

``MMDiTEngine`` stands where ``pipe.transformer`` stands in the reference (src/SD3NaturalInference.py:111-114,
210-213): ``engine(hidden_states=..., timestep=..., encoder_hidden_states=..., pooled_projections=...,
return_dict=False)[0]`` has the call shape of ``diffusers.SD3Transformer2DModel``.  PyTorch only provides device
memory and the stream.
"""
from __future__ import annotations

import ctypes as C
import os
import math
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from ._lib import lib, check, ptr, stream_ptr

FP8 = 1
SD3_MEDIUM = dict(layers=24, heads=24, joint_dim=4096, pooled_dim=2048, in_ch=16)


def param_layout(layers: int, heads: int, joint_dim: int, pooled_dim: int, in_ch: int = 16) -> List[Tuple[str, Tuple[int, ...]]]:
    """Flat parameter order of ``natinf_mmdit_load`` with the diffusers state-dict names (``pos_embed.pos_embed`` is
    cropped to the grid by ``flatten_state_dict``; its shape here is the checkpoint's trailing dimension only)."""
    D = heads * 64
    out: List[Tuple[str, Tuple[int, ...]]] = [
        ("pos_embed.pos_embed", (-1, D)), ("pos_embed.proj.weight", (D, in_ch, 2, 2)), ("pos_embed.proj.bias", (D,))]

    def lin(name, o, i):
        out.extend([(name + ".weight", (o, i)), (name + ".bias", (o,))])
    lin("time_text_embed.timestep_embedder.linear_1", D, 256); lin("time_text_embed.timestep_embedder.linear_2", D, D)
    lin("time_text_embed.text_embedder.linear_1", D, pooled_dim); lin("time_text_embed.text_embedder.linear_2", D, D)
    lin("context_embedder", D, joint_dim)
    for i in range(layers):
        p = f"transformer_blocks.{i}."
        last = i == layers - 1
        lin(p + "norm1.linear", 6 * D, D); lin(p + "norm1_context.linear", (2 if last else 6) * D, D)
        for n in ("to_q", "to_k", "to_v", "add_k_proj", "add_v_proj", "add_q_proj", "to_out.0"):
            lin(p + "attn." + n, D, D)
        if not last:
            lin(p + "attn.to_add_out", D, D)
        lin(p + "ff.net.0.proj", 4 * D, D); lin(p + "ff.net.2", D, 4 * D)
        if not last:
            lin(p + "ff_context.net.0.proj", 4 * D, D); lin(p + "ff_context.net.2", D, 4 * D)
    lin("norm_out.linear", 2 * D, D); lin("proj_out", 4 * in_ch, D)
    return out


def crop_pos_embed(table: torch.Tensor, grid: int) -> torch.Tensor:
    """centre ``grid x grid`` window of the checkpoint's [1, S*S, D] position table -> [grid*grid, D]."""
    S = int(round(math.sqrt(table.shape[-2])))
    top = (S - grid) // 2
    return table.reshape(S, S, -1)[top:top + grid, top:top + grid].reshape(grid * grid, -1)


def flatten_state_dict(sd: Dict[str, torch.Tensor], grid: int, layers: int, heads: int, joint_dim: int, pooled_dim: int,
                       in_ch: int = 16) -> torch.Tensor:
    parts = []
    for name, shape in param_layout(layers, heads, joint_dim, pooled_dim, in_ch):
        t = sd[name].detach().to(torch.float32)
        if name == "pos_embed.pos_embed":
            t = crop_pos_embed(t, grid)
        elif tuple(t.shape) != shape:
            raise ValueError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
        parts.append(t.reshape(-1))
    return torch.cat(parts)


class MMDiTEngine:
    def __init__(self, flat_params: torch.Tensor, max_batch: int, grid: int = 64, ctx_tokens: int = 333, layers: int = 24,
                 heads: int = 24, joint_dim: int = 4096, pooled_dim: int = 2048, in_ch: int = 16, device="cuda:0", fp8: bool = False,
                 stream16: Optional[bool] = None):
        _lib.require_gpu()
        if min(layers, heads, joint_dim, pooled_dim, in_ch, grid, ctx_tokens) <= 0 or heads > 24 or joint_dim % 8 or pooled_dim % 8 \
                or in_ch % 2 or (grid * grid) % 8:
            raise ValueError("unsupported MMDiT configuration (see include/natinf_mmdit.h)")
        if fp8 and heads % 2:
            raise ValueError("fp8 mode needs an even head count")
        self.device = torch.device(device)
        self.max_batch, self.grid, self.ctx_tokens, self.joint_dim, self.pooled_dim, self.in_ch = int(max_batch), grid, ctx_tokens, joint_dim, pooled_dim, in_ch
        self._h = C.c_void_p()
        # ``stream16`` (None = the library's default; the environment variable NATINF_MMDIT_STREAM16 = 0 / 1 overrides that default for A/B runs): the image
        # tokens' residual stream in IEEE half instead of fp32 (include/natinf_mmdit.h, natinf_set_mmdit_stream16: read when the engine is created)
        if stream16 is None and os.environ.get("NATINF_MMDIT_STREAM16") is not None:
            stream16 = bool(int(os.environ["NATINF_MMDIT_STREAM16"]))
        if stream16 is not None:
            check(lib.natinf_set_mmdit_stream16(int(bool(stream16))), "natinf_set_mmdit_stream16")
        try:
            check(lib.natinf_mmdit_create(C.byref(self._h), layers, heads, joint_dim, pooled_dim, in_ch, grid, ctx_tokens, FP8 if fp8 else 0), "natinf_mmdit_create")
        finally:
            if stream16 is not None:
                lib.natinf_set_mmdit_stream16(-1)
        n = lib.natinf_mmdit_param_count(self._h)
        if flat_params.numel() != n:
            raise ValueError(f"expected {n} parameters, got {flat_params.numel()}")
        with torch.cuda.device(self.device):
            params = flat_params.to(self.device, torch.float32).contiguous()
            self._packed = torch.empty(lib.natinf_mmdit_packed_bytes(self._h), dtype=torch.uint8, device=self.device)
            check(lib.natinf_mmdit_load(self._h, ptr(params), n, ptr(self._packed), self._packed.numel(), stream_ptr()), "natinf_mmdit_load")
            torch.cuda.current_stream().synchronize()
            self.workspace_bytes = lib.natinf_mmdit_workspace_bytes(self._h, self.max_batch)
            self._ws = torch.empty(self.workspace_bytes, dtype=torch.uint8, device=self.device)

    def forward(self, latents: torch.Tensor, timestep: torch.Tensor, text: torch.Tensor, pooled: torch.Tensor) -> torch.Tensor:
        S = 2 * self.grid
        if latents.dim() != 4 or tuple(latents.shape[1:]) != (self.in_ch, S, S) or not latents.is_cuda:
            raise ValueError(f"latents must be a CUDA tensor of shape [B,{self.in_ch},{S},{S}]")
        B = latents.shape[0]
        if B > self.max_batch:
            raise ValueError(f"batch {B} exceeds max_batch {self.max_batch}")
        if tuple(text.shape) != (B, self.ctx_tokens, self.joint_dim) or tuple(pooled.shape) != (B, self.pooled_dim) or timestep.numel() != B:
            raise ValueError("timestep [B], encoder_hidden_states [B,ctx_tokens,joint_dim], pooled_projections [B,pooled_dim] expected")
        f = lambda t: t.to(latents.device, torch.float32).contiguous()
        z, ts, tx, pl = f(latents), f(timestep), f(text), f(pooled)
        out = torch.empty_like(z)
        check(lib.natinf_mmdit_forward(self._h, ptr(z), ptr(ts), ptr(tx), ptr(pl), ptr(out), B, ptr(self._ws), self._ws.numel(),
                                       stream_ptr()), "natinf_mmdit_forward")
        return out.to(latents.dtype)

    def __call__(self, hidden_states, timestep, encoder_hidden_states, pooled_projections, return_dict=False, **_):
        """call shape of ``pipe.transformer`` at the reference's call sites; returns a 1-tuple like ``return_dict=False``."""
        return (self.forward(hidden_states, timestep, encoder_hidden_states, pooled_projections),)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib.natinf_mmdit_destroy(h)
            self._h = None


def attention_hd64(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """softmax(q k^T / 8) v for bf16 [B, T, H*64] tensors through ``natinf_attention_hd64_bf16`` (pads T to 128 and
    transposes V on the host side; a utility for tests and benchmarks, the engine feeds the kernel directly)."""
    _lib.require_gpu()
    B, T, D = q.shape
    H = D // 64
    Tp = (T + 127) // 128 * 128
    pad = lambda t, value=0.0: torch.nn.functional.pad(t.to(torch.bfloat16), (0, 0, 0, Tp - T), value=value).contiguous()
    qp, kp = pad(q), pad(k, 25.0)       # (the padded KEY rows hold a large value on purpose: the kernel must mask them, whatever they contain; V^T padding must be finite)
    vT = pad(v).transpose(1, 2).contiguous()
    o = torch.empty_like(qp)
    check(lib.natinf_attention_hd64_bf16(ptr(qp), ptr(kp), D, Tp * D, ptr(vT), ptr(o), D, Tp * D, B, H, Tp, T, 0.125, stream_ptr()),
          "natinf_attention_hd64_bf16")
    return o[:, :T]
