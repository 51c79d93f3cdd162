"""Host wrapper of the gfx950 AutoencoderKL decoder engine (include/natinf_vae.h).

``VAEDecoder`` stands where ``vae.decode`` stands in the reference (src/ValidateNaturalInference.py:231-236: latents divided
by 0.18215, decoded, saved): ``decoder(latents)`` returns images [B, 3, 8r, 8r] like ``vae.decode(latents).sample``.
PyTorch only provides device memory and the stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import torch

from . import _lib
from ._lib import lib, check, ptr, stream_ptr

CH = (512, 512, 256, 128)


def param_layout(latent_ch: int = 4) -> List[Tuple[str, Tuple[int, ...]]]:
    """Flat parameter order of ``natinf_vae_load`` with diffusers' ``Decoder`` state-dict names (prefix ``decoder.`` in a
    full AutoencoderKL checkpoint)."""
    out: List[Tuple[str, Tuple[int, ...]]] = [("conv_in.weight", (512, latent_ch, 3, 3)), ("conv_in.bias", (512,))]
    # (natinf_vae_load takes post_quant_conv.{weight,bias} in front of these: flatten_state_dict adds them)

    def res(p, cin, cout):
        out.extend([(p + "norm1.weight", (cin,)), (p + "norm1.bias", (cin,)), (p + "conv1.weight", (cout, cin, 3, 3)), (p + "conv1.bias", (cout,)),
                    (p + "norm2.weight", (cout,)), (p + "norm2.bias", (cout,)), (p + "conv2.weight", (cout, cout, 3, 3)), (p + "conv2.bias", (cout,))])
        if cin != cout:
            out.extend([(p + "conv_shortcut.weight", (cout, cin, 1, 1)), (p + "conv_shortcut.bias", (cout,))])
    res("mid_block.resnets.0.", 512, 512)
    a = "mid_block.attentions.0."
    out.extend([(a + "group_norm.weight", (512,)), (a + "group_norm.bias", (512,))])
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        out.extend([(a + n + ".weight", (512, 512)), (a + n + ".bias", (512,))])
    res("mid_block.resnets.1.", 512, 512)
    cin = 512
    for i, cout in enumerate(CH):
        for j in range(3):
            res(f"up_blocks.{i}.resnets.{j}.", cin if j == 0 else cout, cout)
        if i < 3:
            out.extend([(f"up_blocks.{i}.upsamplers.0.conv.weight", (cout, cout, 3, 3)), (f"up_blocks.{i}.upsamplers.0.conv.bias", (cout,))])
        cin = cout
    out.extend([("conv_norm_out.weight", (128,)), ("conv_norm_out.bias", (128,)), ("conv_out.weight", (3, 128, 3, 3)), ("conv_out.bias", (3,))])
    return out


_OLD_ATTN_KEYS = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def _modernise_attention_keys(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Checkpoints written before diffusers renamed its attention block (e.g. the 2022 ``sd-vae-ft-*`` repos the
    reference's ``vae_path`` points at) store ``...attentions.0.{query,key,value,proj_attn}`` -- as Linear [C, C] or as 1x1
    conv [C, C, 1, 1] weights; ``AutoencoderKL.from_pretrained`` maps them to ``to_q / to_k / to_v / to_out.0`` on load.
    Same mapping here, so the raw weights file is accepted."""
    out = {}
    for k, v in sd.items():
        parts = k.split(".")
        if len(parts) >= 4 and parts[-4] == "attentions" and parts[-2] in _OLD_ATTN_KEYS:
            k = ".".join(parts[:-2] + [_OLD_ATTN_KEYS[parts[-2]], parts[-1]])
        if ".attentions." in k and k.endswith(".weight") and v.dim() == 4 and v.shape[2:] == (1, 1):
            v = v[:, :, 0, 0]
        out[k] = v
    return out


def flatten_state_dict(sd: Dict[str, torch.Tensor], latent_ch: int = 4, prefix: str = "") -> torch.Tensor:
    """state dict -> the flat fp32 vector of ``natinf_vae_load``.  ``prefix='decoder.'`` for a whole AutoencoderKL checkpoint,
    whose ``post_quant_conv`` (1x1 on the latents, applied by ``AutoencoderKL.decode``) is picked up too; a bare decoder
    state dict gets the identity there."""
    sd = _modernise_attention_keys(sd)
    if "post_quant_conv.weight" in sd:
        parts = [sd["post_quant_conv.weight"].detach().to(torch.float32).reshape(-1), sd["post_quant_conv.bias"].detach().to(torch.float32).reshape(-1)]
        if parts[0].numel() != latent_ch * latent_ch:
            raise ValueError("post_quant_conv does not match latent_ch")
    else:
        parts = [torch.eye(latent_ch).reshape(-1), torch.zeros(latent_ch)]
    for name, shape in param_layout(latent_ch):
        t = sd[prefix + name]
        if tuple(t.shape) != shape:
            raise ValueError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
        parts.append(t.detach().to(torch.float32).reshape(-1))
    return torch.cat(parts)


class VAEDecoder:
    def __init__(self, flat_params: torch.Tensor, max_batch: int, latent_ch: int = 4, latent_res: int = 32, device="cuda:0"):
        _lib.require_gpu()
        if not (1 <= latent_ch <= 64) or latent_res not in (8, 16, 32, 64, 128):
            raise ValueError("latent_ch in 1..64, latent_res one of 8, 16, 32, 64, 128")
        self.device = torch.device(device)
        self.max_batch, self.latent_ch, self.latent_res = int(max_batch), latent_ch, latent_res
        self._h = C.c_void_p()
        check(lib.natinf_vae_create(C.byref(self._h), latent_ch, latent_res), "natinf_vae_create")
        n = lib.natinf_vae_param_count(self._h)
        if flat_params.numel() != n:
            raise ValueError(f"expected {n} parameters, got {flat_params.numel()}")
        with torch.cuda.device(self.device):
            params = flat_params.to(self.device, torch.float32).contiguous()
            self._packed = torch.empty(lib.natinf_vae_packed_bytes(self._h), dtype=torch.uint8, device=self.device)
            check(lib.natinf_vae_load(self._h, ptr(params), n, ptr(self._packed), self._packed.numel(), stream_ptr()), "natinf_vae_load")
            torch.cuda.current_stream().synchronize()
            self.workspace_bytes = lib.natinf_vae_workspace_bytes(self._h, self.max_batch)
            self._ws = torch.empty(self.workspace_bytes, dtype=torch.uint8, device=self.device)

    def __call__(self, latents: torch.Tensor) -> torch.Tensor:
        r = self.latent_res
        if latents.dim() != 4 or tuple(latents.shape[1:]) != (self.latent_ch, r, r) or not latents.is_cuda:
            raise ValueError(f"latents must be a CUDA tensor of shape [B,{self.latent_ch},{r},{r}]")
        B = latents.shape[0]
        if B > self.max_batch:
            raise ValueError(f"batch {B} exceeds max_batch {self.max_batch}")
        z = latents.to(torch.float32).contiguous()
        out = torch.empty((B, 3, 8 * r, 8 * r), dtype=torch.float32, device=z.device)
        check(lib.natinf_vae_decode(self._h, ptr(z), ptr(out), B, ptr(self._ws), self._ws.numel(), stream_ptr()), "natinf_vae_decode")
        return out.to(latents.dtype)

    decode = __call__

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib.natinf_vae_destroy(h)
            self._h = None
