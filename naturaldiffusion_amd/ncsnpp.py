"""Host wrapper of the gfx950 NCSN++ denoiser engine (include/natinf_ncsnpp.h).

``NCSNppEngine`` is what the reference's ``mutils.create_model(config)`` + ``restore_checkpoint`` +
``ema.copy_to`` produce (src/CIFAR10NaturalInference.py:258-265), as one object whose ``__call__(x,
labels)`` replaces ``model(x, labels)`` (deps/score_sde_pytorch/models/utils.py:118-123).  PyTorch only
provides device memory and the stream; every FLOP runs in libnatinf.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import torch

from . import _lib
from ._lib import lib, check, ptr, stream_ptr

KEEP_ACTIVATIONS = 1
DDPM = 2            # NATINF_NCSNPP_DDPM: the `ddpm` network (models/ddpm.py) instead of NCSN++ / DDPM++
ARCH_FLAGS = {"ncsnpp": 0, "ddpm": DDPM}


def module_table(handle=None, arch: str = "ncsnpp") -> List[Tuple[int, str, int, int, int, int, int, int]]:
    """(idx, kind, cin, cout, up, down, res, param_offset) per ``all_modules`` entry, from the engine's plan."""
    own = handle is None
    if own:
        h = C.c_void_p()
        check(lib.natinf_ncsnpp_create(C.byref(h), ARCH_FLAGS[arch]), "natinf_ncsnpp_create")
        handle = h
    buf = C.create_string_buffer(1 << 14)
    n = lib.natinf_ncsnpp_describe(handle, buf, len(buf))
    if n < 0:
        check(n, "natinf_ncsnpp_describe")
    if own:
        lib.natinf_ncsnpp_destroy(handle)
    rows = []
    for line in buf.value.decode().strip().split("\n"):
        f = line.split()
        rows.append((int(f[0]), f[1], *map(int, f[2:])))
    return rows


def param_layout(nf: int = 128, arch: str = "ncsnpp") -> List[Tuple[str, Tuple[int, ...]]]:
    """Flat parameter order the engine expects: ``all_modules`` order, leaves in registration order
    (= ``model.parameters()`` order = EMA ``shadow_params`` order, ema.py:28-29).  ``nf`` is ``config.model.nf``
    (configs/vp/cifar10_ddpmpp_continuous.py:47): the engine is built for 128; other widths only serve the host-side
    loaders (checkpoint-format tests on a small reference-written pickle)."""
    out: List[Tuple[str, Tuple[int, ...]]] = []
    sc = lambda c: c if c == 3 else c * nf // 128           # image channels stay 3; every feature width is a multiple of nf
    TEMB = 4 * nf
    for idx, kind, cin, cout, up, down, res, _ in module_table(arch=arch):
        cin, cout = sc(cin), sc(cout)
        p = f"all_modules.{idx}."
        if kind == "lin":
            out += [(p + "weight", (cout, cin)), (p + "bias", (cout,))]
        elif kind == "conv":
            out += [(p + "weight", (cout, cin, 3, 3)), (p + "bias", (cout,))]
        elif kind == "gn":
            out += [(p + "weight", (cin,)), (p + "bias", (cin,))]
        elif kind == "attn":
            out += [(p + "GroupNorm_0.weight", (cin,)), (p + "GroupNorm_0.bias", (cin,))]
            for i in range(4):
                out += [(p + f"NIN_{i}.W", (cin, cin)), (p + f"NIN_{i}.b", (cin,))]
        elif kind == "res":
            out += [(p + "GroupNorm_0.weight", (cin,)), (p + "GroupNorm_0.bias", (cin,)),
                    (p + "Conv_0.weight", (cout, cin, 3, 3)), (p + "Conv_0.bias", (cout,)),
                    (p + "Dense_0.weight", (cout, TEMB)), (p + "Dense_0.bias", (cout,)),
                    (p + "GroupNorm_1.weight", (cout,)), (p + "GroupNorm_1.bias", (cout,)),
                    (p + "Conv_1.weight", (cout, cout, 3, 3)), (p + "Conv_1.bias", (cout,))]
            if arch == "ddpm":
                if cin != cout:                             # ResnetBlockDDPM's NIN shortcut (layers.py:632-636): W is [in][out]
                    out += [(p + "NIN_0.W", (cin, cout)), (p + "NIN_0.b", (cout,))]
            elif cin != cout or up or down:
                out += [(p + "Conv_2.weight", (cout, cin, 1, 1)), (p + "Conv_2.bias", (cout,))]
        elif kind in ("down", "up"):                        # `ddpm` Downsample / Upsample with resamp_with_conv (layers.py:586-612)
            out += [(p + "Conv_0.weight", (cout, cin, 3, 3)), (p + "Conv_0.bias", (cout,))]
        else:
            raise RuntimeError(f"unknown module kind {kind}")
    return out


def flatten_state_dict(sd: Dict[str, torch.Tensor], arch: str = "ncsnpp") -> torch.Tensor:
    """name -> tensor dict (reference keys, with or without the DataParallel ``module.`` prefix,
    models/utils.py:93) -> one fp32 CPU vector in engine order.  Shapes are checked."""
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    parts = []
    for name, shape in param_layout(arch=arch):
        if name not in sd:
            raise KeyError(f"checkpoint lacks {name}")
        t = sd[name]
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: shape {tuple(t.shape)} != expected {shape}")
        parts.append(t.detach().to("cpu", torch.float32).reshape(-1))
    return torch.cat(parts)


def flatten_ema(shadow_params: List[torch.Tensor], nf: int = 128, arch: str = "ncsnpp") -> torch.Tensor:
    """EMA list of the score_sde checkpoint (``state['ema']['shadow_params']``, ema.py:91-97)."""
    layout = param_layout(nf, arch)
    if len(shadow_params) != len(layout):
        raise ValueError(f"EMA list has {len(shadow_params)} tensors, the model has {len(layout)} parameters")
    parts = []
    for (name, shape), t in zip(layout, shadow_params):
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"EMA entry for {name}: shape {tuple(t.shape)} != expected {shape}")
        parts.append(t.detach().to("cpu", torch.float32).reshape(-1))
    return torch.cat(parts)


def load_score_sde_checkpoint(path: str, nf: int = 128, arch: str = "ncsnpp") -> torch.Tensor:
    """``restore_checkpoint`` + ``ema.copy_to`` (deps/score_sde_pytorch/utils.py:7-19, ema.py:53-64):
    the weights the reference samples with are the EMA shadow parameters.  The pickle's ``model`` entry (DataParallel
    ``module.`` keys plus the ``sigmas`` buffer) is only used to cross-check the names and shapes of the EMA list."""
    state = torch.load(path, map_location="cpu", weights_only=False)
    for key in ("model", "ema", "step"):
        if key not in state:
            raise KeyError(f"{path}: not a score_sde checkpoint (no '{key}' entry; utils.py:22-29)")
    shadow = list(state["ema"]["shadow_params"])
    model_sd = {(k[7:] if k.startswith("module.") else k): v for k, v in state["model"].items()}
    for (name, shape), t in zip(param_layout(nf, arch), shadow):
        if name in model_sd and tuple(model_sd[name].shape) != tuple(t.shape):
            raise ValueError(f"{path}: EMA entry for {name} has shape {tuple(t.shape)}, the model's is {tuple(model_sd[name].shape)}")
    return flatten_ema(shadow, nf, arch)


class NCSNppEngine:
    def __init__(self, flat_params: torch.Tensor, max_batch: int, device="cuda:0", keep_activations: bool = False, arch: str = "ncsnpp"):
        """``arch``: "ncsnpp" (configs/vp/cifar10_ddpmpp_continuous.py, what the reference script imports) or "ddpm"
        (configs/vp/ddpm/cifar10_continuous.py, the checkpoint its docstring names)."""
        _lib.require_gpu()
        self.device = torch.device(device)
        self.max_batch = int(max_batch)
        self.arch = arch
        self._h = C.c_void_p()
        self._flags = (KEEP_ACTIVATIONS if keep_activations else 0) | ARCH_FLAGS[arch]
        check(lib.natinf_ncsnpp_create(C.byref(self._h), self._flags), "natinf_ncsnpp_create")
        n = lib.natinf_ncsnpp_handle_param_count(self._h)
        if flat_params.numel() != n:
            raise ValueError(f"expected {n} parameters, got {flat_params.numel()}")
        with torch.cuda.device(self.device):
            params = flat_params.to(self.device, torch.float32).contiguous()
            self._packed = torch.empty(lib.natinf_ncsnpp_handle_packed_bytes(self._h), dtype=torch.uint8, device=self.device)
            check(lib.natinf_ncsnpp_load(self._h, ptr(params), n, ptr(self._packed), self._packed.numel(), stream_ptr()),
                  "natinf_ncsnpp_load")
            torch.cuda.current_stream().synchronize()      # params may be freed after this
            ws = lib.natinf_ncsnpp_workspace_bytes(self._h, self.max_batch)
            self._ws = torch.empty(ws, dtype=torch.uint8, device=self.device)
        self.workspace_bytes = ws

    def clone(self, max_batch: int = None) -> "NCSNppEngine":
        """A second handle of the same network for a second HIP stream: its own launch plan and workspace, the SAME packed weights
        (``natinf_ncsnpp_share``: read-only during forwards, so one 124 MB copy serves every lane)."""
        other = object.__new__(NCSNppEngine)
        other.device, other.arch = self.device, self.arch
        other.max_batch = int(max_batch or self.max_batch)
        other._h = C.c_void_p()
        check(lib.natinf_ncsnpp_create(C.byref(other._h), self._flags), "natinf_ncsnpp_create")
        check(lib.natinf_ncsnpp_share(other._h, self._h), "natinf_ncsnpp_share")
        other._flags = self._flags
        other._packed = self._packed                       # keeps the shared buffer alive as long as any handle lives
        with torch.cuda.device(self.device):
            ws = lib.natinf_ncsnpp_workspace_bytes(other._h, other.max_batch)
            other._ws = torch.empty(ws, dtype=torch.uint8, device=self.device)
        other.workspace_bytes = ws
        return other

    def __call__(self, x: torch.Tensor, labels: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        if x.dtype != torch.float32 or x.dim() != 4 or tuple(x.shape[1:]) != (3, 32, 32) or not x.is_cuda:
            raise ValueError("x must be a CUDA fp32 tensor of shape [B,3,32,32]")
        B = x.shape[0]
        if B > self.max_batch:
            raise ValueError(f"batch {B} exceeds max_batch {self.max_batch}")
        x = x.contiguous()
        labels = labels.to(x.device, torch.float32).contiguous()
        if labels.numel() != B:
            raise ValueError("labels must have one entry per sample")
        if out is None:
            out = torch.empty_like(x)
        check(lib.natinf_ncsnpp_forward(self._h, ptr(x), ptr(labels), ptr(out), B, ptr(self._ws), self._ws.numel(),
                                        stream_ptr()), "natinf_ncsnpp_forward")
        return out

    def profile(self, enable: bool) -> None:
        check(lib.natinf_ncsnpp_profile(self._h, 1 if enable else 0), "natinf_ncsnpp_profile")

    def profile_read(self):
        """-> {"gemm": (ms, launches), "other": (...), "conv_gn": (...), "conv_gn8": (...)} since the last read (synchronises)."""
        ms = (C.c_double * 4)()
        n = (C.c_int64 * 4)()
        check(lib.natinf_ncsnpp_profile_read(self._h, ms, n), "natinf_ncsnpp_profile_read")
        return {"gemm": (ms[0], n[0]), "other": (ms[1], n[1]), "conv_gn": (ms[2], n[2]), "conv_gn8": (ms[3], n[3])}

    def describe_gemms(self, B: int):
        """[(M, N, K0, K1, taps, batch, "variant/eN")] of every matmul-shaped launch of a forward at batch B, in launch order."""
        buf = C.create_string_buffer(1 << 16)
        n = lib.natinf_ncsnpp_describe_gemms(self._h, B, buf, len(buf))
        if n < 0:
            check(n, "natinf_ncsnpp_describe_gemms")
        rows = []
        for line in buf.value.decode().strip().split("\n"):
            f = line.split()
            rows.append((*map(int, f[:6]), f[6]))
        return rows

    def tap(self, module_idx: int, shape) -> torch.Tensor:
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        check(lib.natinf_ncsnpp_debug_tap(self._h, module_idx, ptr(out), out.numel(), stream_ptr()), "natinf_ncsnpp_debug_tap")
        return out

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib.natinf_ncsnpp_destroy(h)
            self._h = None
