"""Host wrapper of the gfx950 FID Inception-V3 pool3 engine (include/natinf_inception.h).

``InceptionEngine`` stands where ``pytorch_fid.inception.InceptionV3([BLOCK_INDEX_BY_DIM[2048]])`` stands in the reference's FID
epilogue (src/CIFAR10NaturalInference.py:44-86): ``engine(batch)`` takes the uint8 HWC images the samplers produce (or the float NCHW
batch in [0, 1] the reference builds from them) and returns the pool3 features [n, 2048].  PyTorch only provides device memory and
the stream; the weights (``pt_inception-2015-12-05-6726825d.pth``, a download) are loaded by ``load_fid_inception_weights``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import torch

from . import _lib
from ._lib import lib, check, ptr, stream_ptr

U8_HWC, F32_CHW = 0, 1


def conv_specs() -> List[Tuple[str, int, int, Tuple[int, int], int, Tuple[int, int]]]:
    """(name, cin, cout, (kh, kw), stride, (ph, pw)) of every BasicConv2d on the pool3 path, in torchvision registration order --
    the parameter order of ``natinf_inception_load``."""
    s: List = [("Conv2d_1a_3x3", 3, 32, (3, 3), 2, (0, 0)), ("Conv2d_2a_3x3", 32, 32, (3, 3), 1, (0, 0)), ("Conv2d_2b_3x3", 32, 64, (3, 3), 1, (1, 1)),
               ("Conv2d_3b_1x1", 64, 80, (1, 1), 1, (0, 0)), ("Conv2d_4a_3x3", 80, 192, (3, 3), 1, (0, 0))]
    one = lambda n, i, o: (n, i, o, (1, 1), 1, (0, 0))
    for n, cin, pf in (("Mixed_5b", 192, 32), ("Mixed_5c", 256, 64), ("Mixed_5d", 288, 64)):
        s += [one(f"{n}.branch1x1", cin, 64), one(f"{n}.branch5x5_1", cin, 48), (f"{n}.branch5x5_2", 48, 64, (5, 5), 1, (2, 2)),
              one(f"{n}.branch3x3dbl_1", cin, 64), (f"{n}.branch3x3dbl_2", 64, 96, (3, 3), 1, (1, 1)), (f"{n}.branch3x3dbl_3", 96, 96, (3, 3), 1, (1, 1)),
              one(f"{n}.branch_pool", cin, pf)]
    n = "Mixed_6a"
    s += [(f"{n}.branch3x3", 288, 384, (3, 3), 2, (0, 0)), one(f"{n}.branch3x3dbl_1", 288, 64), (f"{n}.branch3x3dbl_2", 64, 96, (3, 3), 1, (1, 1)),
          (f"{n}.branch3x3dbl_3", 96, 96, (3, 3), 2, (0, 0))]
    h7, v7 = ((1, 7), 1, (0, 3)), ((7, 1), 1, (3, 0))
    for n, c7 in (("Mixed_6b", 128), ("Mixed_6c", 160), ("Mixed_6d", 160), ("Mixed_6e", 192)):
        s += [one(f"{n}.branch1x1", 768, 192), one(f"{n}.branch7x7_1", 768, c7), (f"{n}.branch7x7_2", c7, c7, *h7), (f"{n}.branch7x7_3", c7, 192, *v7),
              one(f"{n}.branch7x7dbl_1", 768, c7), (f"{n}.branch7x7dbl_2", c7, c7, *v7), (f"{n}.branch7x7dbl_3", c7, c7, *h7),
              (f"{n}.branch7x7dbl_4", c7, c7, *v7), (f"{n}.branch7x7dbl_5", c7, 192, *h7), one(f"{n}.branch_pool", 768, 192)]
    n = "Mixed_7a"
    s += [one(f"{n}.branch3x3_1", 768, 192), (f"{n}.branch3x3_2", 192, 320, (3, 3), 2, (0, 0)), one(f"{n}.branch7x7x3_1", 768, 192),
          (f"{n}.branch7x7x3_2", 192, 192, *h7), (f"{n}.branch7x7x3_3", 192, 192, *v7), (f"{n}.branch7x7x3_4", 192, 192, (3, 3), 2, (0, 0))]
    h3, v3 = ((1, 3), 1, (0, 1)), ((3, 1), 1, (1, 0))
    for n, cin in (("Mixed_7b", 1280), ("Mixed_7c", 2048)):
        s += [one(f"{n}.branch1x1", cin, 320), one(f"{n}.branch3x3_1", cin, 384), (f"{n}.branch3x3_2a", 384, 384, *h3), (f"{n}.branch3x3_2b", 384, 384, *v3),
              one(f"{n}.branch3x3dbl_1", cin, 448), (f"{n}.branch3x3dbl_2", 448, 384, (3, 3), 1, (1, 1)), (f"{n}.branch3x3dbl_3a", 384, 384, *h3),
              (f"{n}.branch3x3dbl_3b", 384, 384, *v3), one(f"{n}.branch_pool", cin, 192)]
    return s


def param_layout() -> List[Tuple[str, Tuple[int, ...]]]:
    """state-dict names (torchvision ``Inception3`` / the pt_inception checkpoint) and shapes in ``natinf_inception_load`` order."""
    out = []
    for n, cin, cout, (kh, kw), _, _ in conv_specs():
        out += [(f"{n}.conv.weight", (cout, cin, kh, kw)), (f"{n}.bn.weight", (cout,)), (f"{n}.bn.bias", (cout,)),
                (f"{n}.bn.running_mean", (cout,)), (f"{n}.bn.running_var", (cout,))]
    return out


# pytorch_fid wraps the torchvision modules in nn.Sequential blocks: its own state_dict says blocks.<i>.<j>.* for these modules
_FID_BLOCKS = {"blocks.0.0": "Conv2d_1a_3x3", "blocks.0.1": "Conv2d_2a_3x3", "blocks.0.2": "Conv2d_2b_3x3", "blocks.1.0": "Conv2d_3b_1x1",
               "blocks.1.1": "Conv2d_4a_3x3", "blocks.2.0": "Mixed_5b", "blocks.2.1": "Mixed_5c", "blocks.2.2": "Mixed_5d", "blocks.2.3": "Mixed_6a",
               "blocks.2.4": "Mixed_6b", "blocks.2.5": "Mixed_6c", "blocks.2.6": "Mixed_6d", "blocks.2.7": "Mixed_6e", "blocks.3.0": "Mixed_7a",
               "blocks.3.1": "Mixed_7b", "blocks.3.2": "Mixed_7c"}


def flatten_state_dict(sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """torchvision-named (the checkpoint file) or pytorch_fid-named (``InceptionV3(...).state_dict()``) weights -> the flat fp32 vector."""
    norm = {}
    for k, v in sd.items():
        parts = k.split(".")
        head = ".".join(parts[:3])
        if head in _FID_BLOCKS:
            k = ".".join([_FID_BLOCKS[head]] + parts[3:])
        norm[k] = v
    parts = []
    for name, shape in param_layout():
        if name not in norm:
            raise KeyError(f"Inception weights: {name} missing")
        t = norm[name].detach().to(torch.float32)
        if tuple(t.shape) != shape:
            raise ValueError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
        parts.append(t.reshape(-1))
    return torch.cat(parts)


def load_fid_inception_weights(path) -> torch.Tensor:
    """``pt_inception-2015-12-05-6726825d.pth`` (what pytorch_fid downloads) -> flat parameters."""
    return flatten_state_dict(torch.load(str(path), map_location="cpu", weights_only=True))


class InceptionEngine:
    def __init__(self, flat_params: torch.Tensor, max_batch: int = 50, in_hw: Tuple[int, int] = (32, 32), device="cuda:0"):
        _lib.require_gpu()
        self.device = torch.device(device)
        self.max_batch, self.in_hw = int(max_batch), (int(in_hw[0]), int(in_hw[1]))
        self._h = C.c_void_p()
        check(lib.natinf_inception_create(C.byref(self._h), self.in_hw[0], self.in_hw[1]), "natinf_inception_create")
        n = lib.natinf_inception_param_count(self._h)
        if flat_params.numel() != n:
            raise ValueError(f"expected {n} parameters, got {flat_params.numel()}")
        with torch.cuda.device(self.device):
            params = flat_params.to(self.device, torch.float32).contiguous()
            self._packed = torch.empty(lib.natinf_inception_packed_bytes(self._h), dtype=torch.uint8, device=self.device)
            check(lib.natinf_inception_load(self._h, ptr(params), n, ptr(self._packed), self._packed.numel(), stream_ptr()), "natinf_inception_load")
            torch.cuda.current_stream().synchronize()
            self._ws = torch.empty(lib.natinf_inception_workspace_bytes(self._h, self.max_batch), dtype=torch.uint8, device=self.device)

    def __call__(self, images: torch.Tensor) -> torch.Tensor:
        """uint8 [B, H, W, 3] or float [B, 3, H, W] in [0, 1]  ->  pool3 features fp32 [B, 2048] (any B: walked in max_batch chunks)."""
        H, W = self.in_hw
        if images.dtype == torch.uint8:
            if images.dim() != 4 or tuple(images.shape[1:]) != (H, W, 3):
                raise ValueError(f"uint8 images must be [B,{H},{W},3]")
            kind, x = U8_HWC, images.to(self.device).contiguous()
        else:
            if images.dim() != 4 or tuple(images.shape[1:]) != (3, H, W):
                raise ValueError(f"float images must be [B,3,{H},{W}]")
            kind, x = F32_CHW, images.to(self.device, torch.float32).contiguous()
        out = torch.empty(x.shape[0], 2048, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            for i in range(0, x.shape[0], self.max_batch):
                xb = x[i:i + self.max_batch]
                check(lib.natinf_inception_forward(self._h, ptr(xb), kind, ptr(out[i:i + self.max_batch]), xb.shape[0], ptr(self._ws), self._ws.numel(),
                                                   stream_ptr()), "natinf_inception_forward")
        return out

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib.natinf_inception_destroy(h)
            self._h = None
