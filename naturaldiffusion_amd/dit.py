"""Host wrapper of the gfx950 DiT denoiser engine (include/natinf_dit.h).

``DiTEngine`` stands where ``DiT_models['DiT-XL/2'](input_size=32, num_classes=1000)`` + ``load_state_dict``
stand in the reference (src/ValidateNaturalInference.py:150-154); ``engine(z, t, y)`` replaces
``model.forward(z, t, y)`` (deps/DiT/models.py:237-253).  PyTorch only provides device memory and the stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import torch

from . import _lib
from ._lib import lib, check, ptr, stream_ptr

UNFUSED_ATTENTION = 1
XL2 = dict(depth=28, hidden=1152, heads=16)          # deps/DiT/models.py:333-334


def param_layout(depth: int, hidden: int) -> List[Tuple[str, Tuple[int, ...]]]:
    """Flat parameter order of ``natinf_dit_load`` with the reference's state-dict names and shapes."""
    D = hidden
    out: List[Tuple[str, Tuple[int, ...]]] = [
        ("pos_embed", (1, 256, D)), ("x_embedder.proj.weight", (D, 4, 2, 2)), ("x_embedder.proj.bias", (D,)),
        ("t_embedder.mlp.0.weight", (D, 256)), ("t_embedder.mlp.0.bias", (D,)),
        ("t_embedder.mlp.2.weight", (D, D)), ("t_embedder.mlp.2.bias", (D,)),
        ("y_embedder.embedding_table.weight", (1001, D))]
    for i in range(depth):
        p = f"blocks.{i}."
        out += [(p + "attn.qkv.weight", (3 * D, D)), (p + "attn.qkv.bias", (3 * D,)),
                (p + "attn.proj.weight", (D, D)), (p + "attn.proj.bias", (D,)),
                (p + "mlp.fc1.weight", (4 * D, D)), (p + "mlp.fc1.bias", (4 * D,)),
                (p + "mlp.fc2.weight", (D, 4 * D)), (p + "mlp.fc2.bias", (D,)),
                (p + "adaLN_modulation.1.weight", (6 * D, D)), (p + "adaLN_modulation.1.bias", (6 * D,))]
    out += [("final_layer.linear.weight", (32, D)), ("final_layer.linear.bias", (32,)),
            ("final_layer.adaLN_modulation.1.weight", (2 * D, D)), ("final_layer.adaLN_modulation.1.bias", (2 * D,))]
    return out


def flatten_state_dict(sd: Dict[str, torch.Tensor], depth: int, hidden: int) -> torch.Tensor:
    """state_dict of the reference's DiT (e.g. ``DiT-XL-2-256x256.pt``) -> the flat fp32 vector of ``natinf_dit_load``."""
    parts = []
    for name, shape in param_layout(depth, hidden):
        t = sd[name]
        if tuple(t.shape) != shape:
            raise ValueError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
        parts.append(t.detach().to(torch.float32).reshape(-1))
    return torch.cat(parts)


class DiTEngine:
    def __init__(self, flat_params: torch.Tensor, max_batch: int, depth: int = 28, hidden: int = 1152, heads: int = 16,
                 device="cuda:0", unfused_attention: bool = False, stream16=None):
        _lib.require_gpu()
        if depth <= 0 or hidden <= 0 or heads <= 0 or hidden % 64 or hidden > 1536 or hidden % heads or (hidden // heads) % 8:
            raise ValueError("hidden must be a multiple of 64 (<= 1536) and of heads, head_dim a multiple of 8")
        self.device = torch.device(device)
        self.max_batch = int(max_batch)
        self._h = C.c_void_p()
        # ``stream16`` (None = the library's default; NATINF_DIT_STREAM16 = 0 / 1 in the environment overrides that default for A/B runs): the residual stream in
        # IEEE half instead of fp32 (include/natinf_dit.h, natinf_set_dit_stream16: read when the engine is created)
        import os
        if stream16 is None and os.environ.get("NATINF_DIT_STREAM16") is not None:
            stream16 = bool(int(os.environ["NATINF_DIT_STREAM16"]))
        if stream16 is not None:
            check(lib.natinf_set_dit_stream16(int(bool(stream16))), "natinf_set_dit_stream16")
        try:
            check(lib.natinf_dit_create(C.byref(self._h), depth, hidden, heads, UNFUSED_ATTENTION if unfused_attention else 0), "natinf_dit_create")
        finally:
            if stream16 is not None:
                lib.natinf_set_dit_stream16(-1)
        n = lib.natinf_dit_param_count(self._h)
        if flat_params.numel() != n:
            raise ValueError(f"expected {n} parameters, got {flat_params.numel()}")
        with torch.cuda.device(self.device):
            params = flat_params.to(self.device, torch.float32).contiguous()
            self._packed = torch.empty(lib.natinf_dit_packed_bytes(self._h), dtype=torch.uint8, device=self.device)
            check(lib.natinf_dit_load(self._h, ptr(params), n, ptr(self._packed), self._packed.numel(), stream_ptr()),
                  "natinf_dit_load")
            torch.cuda.current_stream().synchronize()
            self.workspace_bytes = lib.natinf_dit_workspace_bytes(self._h, self.max_batch)
            self._ws = torch.empty(self.workspace_bytes, dtype=torch.uint8, device=self.device)

    def __call__(self, z: torch.Tensor, t: torch.Tensor, y: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        if z.dtype != torch.float32 or z.dim() != 4 or tuple(z.shape[1:]) != (4, 32, 32) or not z.is_cuda:
            raise ValueError("z must be a CUDA fp32 tensor of shape [B,4,32,32]")
        B = z.shape[0]
        if B > self.max_batch:
            raise ValueError(f"batch {B} exceeds max_batch {self.max_batch}")
        z = z.contiguous()
        t = t.to(z.device, torch.float32).contiguous()
        y = y.to(z.device, torch.int32).contiguous()
        if t.numel() != B or y.numel() != B:
            raise ValueError("t and y must have one entry per sample")
        if out is None:
            out = torch.empty((B, 8, 32, 32), dtype=torch.float32, device=z.device)
        check(lib.natinf_dit_forward(self._h, ptr(z), ptr(t), ptr(y), ptr(out), B, ptr(self._ws), self._ws.numel(),
                                     stream_ptr()), "natinf_dit_forward")
        return out

    forward = __call__

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib.natinf_dit_destroy(h)
            self._h = None
