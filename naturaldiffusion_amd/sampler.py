"""Stateful Natural Inference samplers: one fused ``ni_step`` launch per sampling step.

Each class owns the history slab(s) in HBM ([slot][E], one coalesced stream per history row) and the
device-resident sparse coefficient rows, and mirrors one of the reference's three loop bodies:

* :class:`CifarNI`    -- src/CIFAR10NaturalInference.py:292-304 (fp64 history)
* :class:`ValidateNI` -- src/ValidateNaturalInference.py:349-366 (fp32 products, fp64 accumulate)
* :class:`SD3NI`      -- src/SD3NaturalInference.py:201-221 and :105-129 (fp16 chain)

Host code only sequences launches; all arithmetic is in libnatinf.so (include/natinf.h).
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ptr, stream_ptr
from .coeff import SparseRows


def vp_std_f32(t: float, beta_0: float = 0.1, beta_1: float = 20.0) -> float:
    """sigma(t) of the VP SDE exactly as ``score_fn`` evaluates it (sde_lib.py:141-145 on the fp32
    vector ``t*ones``): host-side scalar schedule, fp32 torch ops on CPU."""
    vt = torch.ones(1, dtype=torch.float32) * t
    lmc = -0.25 * vt ** 2 * (beta_1 - beta_0) - 0.5 * vt * beta_0
    return float(torch.sqrt(1.0 - torch.exp(2.0 * lmc))[0])


class CifarNI:
    """x_{k+1} = fp32(sum_j C[k,j]*x0_j) + fp32(B[k,0])*noise with x0_k = ((-out/std)*sigma^2 + x_k)/alpha."""

    def __init__(self, C: np.ndarray, B: np.ndarray, node: np.ndarray, n_elem: int, device="cuda:0",
                 dense: bool = False, fast_f32: bool = False, stds=None):
        _lib.require_gpu()
        if n_elem % 4:
            raise ValueError("element count must be a multiple of 4")
        self.C, self.B, self.node = (np.asarray(a, np.float64) for a in (C, B, node))
        self.n_step = self.node.shape[0] - 1
        if self.C.shape != (self.n_step, self.n_step):
            raise ValueError("C must be [N,N] with N = len(node_coeff)-1")
        self.E = int(n_elem)
        self.device = torch.device(device)
        self.fast = bool(fast_f32)
        hdt = torch.float32 if self.fast else torch.float64
        self.rows = SparseRows(self.C, lambda k: k + 1, hdt, self.device, dense=dense)
        self.hist = torch.empty((self.n_step, self.E), dtype=hdt, device=self.device)
        self._x = [torch.empty(self.E, dtype=torch.float32, device=self.device) for _ in range(2)]
        # fp32 VP std per step, evaluated on the host like score_fn does; `stds` lets a caller pin them
        # (torch.exp on CPU differs in the last ulp between hosts -- so does the reference's own value)
        self.std = [vp_std_f32(self.node[k, 0]) for k in range(self.n_step)] if stds is None else [float(v) for v in stds]
        self.labels = [float(np.float32(self.node[k, 0]) * np.float32(999)) for k in range(self.n_step)]

    def step(self, k: int, x_k: torch.Tensor, model_out: torch.Tensor, noise: torch.Tensor,
             x_next: Optional[torch.Tensor] = None) -> torch.Tensor:
        if x_next is None:
            x_next = self._x[k & 1]
            if x_next.data_ptr() == x_k.data_ptr():
                x_next = self._x[(k + 1) & 1]
        for t in (x_k, model_out, noise):
            if t.dtype != torch.float32 or t.numel() != self.E or not t.is_contiguous():
                raise ValueError("x_k / model_out / noise must be contiguous fp32 tensors of n_elem elements")
        idx, val, n = self.rows.ptrs(k)
        r = self.rows.rows[k]
        a, s = float(self.node[k, 1]), float(self.node[k, 2])
        b0 = float(np.float32(self.B[k, 0]))
        fn = lib.natinf_step_f32hist if self.fast else lib.natinf_step_f64hist
        check(fn(ptr(x_k), ptr(model_out), ptr(noise), ptr(self.hist), ptr(x_next), idx, val, n, r.diag, k,
                 a, s, self.std[k], b0, self.E, stream_ptr()), "natinf_step_f64hist")
        return x_next

    def run(self, model_fn: Callable, noise: torch.Tensor, return_all: bool = False):
        """``model_fn(x [B,...] fp32, labels [B] fp32) -> out`` (raw network output)."""
        shape, B = noise.shape, noise.shape[0]
        noise = noise.contiguous()
        x, xs = noise, [noise]
        for k in range(self.n_step):
            labels = torch.full((B,), self.labels[k], dtype=torch.float32, device=self.device)
            out = model_fn(x.view(shape), labels)
            x = self.step(k, x.reshape(-1), out.contiguous().reshape(-1), noise.reshape(-1))
            if return_all:
                x = x.clone()
                xs.append(x.view(shape))
        return xs if return_all else x.view(shape)


class ValidateNI:
    """DiT / eps-prediction form with per-step fresh noise and CFG (ValidateNaturalInference.py:311-372)."""

    def __init__(self, C: np.ndarray, B: np.ndarray, node: np.ndarray, c1: np.ndarray, c2: np.ndarray, n_elem: int,
                 device="cuda:0", dense: bool = False):
        _lib.require_gpu()
        if n_elem % 4:
            raise ValueError("element count must be a multiple of 4")
        self.n_step = int(np.asarray(B).shape[0])
        self.node = np.asarray(node, np.float64)
        self.E = int(n_elem)
        self.device = torch.device(device)
        self.rows_c = SparseRows(C, lambda k: k + 1, torch.float32, self.device, dense=dense)
        self.rows_b = SparseRows(B, lambda k: min(k + 2, np.asarray(B).shape[1]), torch.float32, self.device, dense=dense, diag=False)
        self.c1 = [float(np.float32(v)) for v in c1]
        self.c2 = [float(np.float32(v)) for v in c2]
        self.hist_x0 = torch.empty((self.n_step, self.E), dtype=torch.float32, device=self.device)
        self.hist_eps = torch.empty((self.n_step + 1, self.E), dtype=torch.float32, device=self.device)
        self._z = [torch.empty(self.E, dtype=torch.float32, device=self.device) for _ in range(2)]

    def step(self, k: int, z: torch.Tensor, cond: torch.Tensor, uncond: Optional[torch.Tensor], cfg: float,
             sample_elems: Optional[int] = None, eps_sample_stride: Optional[int] = None) -> torch.Tensor:
        se = self.E if sample_elems is None else int(sample_elems)
        st = se if eps_sample_stride is None else int(eps_sample_stride)
        z_next = self._z[k & 1]
        if z_next.data_ptr() == z.data_ptr():
            z_next = self._z[(k + 1) & 1]
        ic, vc, nc = self.rows_c.ptrs(k)
        ib, vb, nb = self.rows_b.ptrs(k)
        check(lib.natinf_step_f32prod(ptr(z), ptr(cond), ptr(uncond), float(cfg), se, st, ptr(self.hist_x0),
                                      ptr(self.hist_eps), ptr(z_next), ic, vc, nc, self.rows_c.rows[k].diag,
                                      ib, vb, nb, k, self.c1[k], self.c2[k], self.E, stream_ptr()), "natinf_step_f32prod")
        return z_next


class SD3NI:
    """Row-normalised fp16 weighted mean + CFG + next flow input (SD3NaturalInference.py:157-168,198-223)."""

    def __init__(self, weights: np.ndarray, sigmas: torch.Tensor, n_elem: int, device="cuda:0", cfg: float = 7.0,
                 dense: bool = False, euler: bool = False):
        _lib.require_gpu()
        if n_elem % 8:
            raise ValueError("element count must be a multiple of 8")
        self.E = int(n_elem)
        self.device = torch.device(device)
        self.cfg = float(cfg)
        self.euler = bool(euler)
        sig = sigmas.detach().to("cpu", torch.float32)
        h = lambda t: float(t.to(torch.float16))                     # 0-d fp32 tensor -> fp16 value, as torch casts it
        if euler:
            # weights w_j = sigma_j - sigma_{j+1} are fp32 0-d tensors: as the FIRST operand of `w * x` eager
            # PyTorch casts them to fp16, but as the SECOND operand of `acc / total` the CPU kernel keeps the
            # fp32 value (original_scalar_value); the row total is the fp32 running sum (SD3...:61-69).
            # This pins eager-CPU semantics (what tests/golden/sd3_form.npz was captured with).  On CUDA the
            # divisor is a 0-d device tensor and the Half functor rounds it to fp16 first: up to 1 fp16 ulp in
            # the mean -- a property of the reference's platform, not of the algorithm; the bit-exact contract
            # of the Euler twin is therefore stated against the CPU run.
            n = sig.numel() - 1
            w32 = [-1 * (sig[i + 1] - sig[i]) for i in range(n)]
            W = np.zeros((n, n))
            tot = []
            for k in range(n):
                acc = 0
                for j in range(k + 1):
                    W[k, j] = h(w32[j])
                    acc = acc + w32[j]
                tot.append(float(acc))
            self.rows = SparseRows(W, lambda k: k + 1, torch.float32, self.device, dense=dense)
            self.totals = tot
        else:
            W = np.asarray(weights, np.float64)
            n = W.shape[0]
            self.rows = SparseRows(W, lambda k: k + 1, torch.float32, self.device, dense=dense)
            self.totals = [r.total for r in self.rows.rows]
        self.n_step = n
        self.sig = [h(sig[k]) for k in range(n + 1)]
        self.oms = [h(1 - sig[k]) for k in range(n + 1)]
        self.hist = torch.empty((n, self.E), dtype=torch.float16, device=self.device)
        self._x = [torch.empty(self.E, dtype=torch.float16, device=self.device) for _ in range(2)]
        self.mean = torch.empty(self.E, dtype=torch.float16, device=self.device)

    def first_input(self, noises: torch.Tensor) -> torch.Tensor:
        """x_0 = sigma_0*noise + (1-sigma_0)*0 in fp16 ops (SD3...:207-209 with an empty history)."""
        n = noises.contiguous().reshape(-1)
        if self.euler:
            return n.clone()                                         # SD3...:94: curr_outputs = deepcopy(noises)
        out = self._x[1]
        check(lib.natinf_flow_input_f16(ptr(n), None, ptr(out), self.sig[0], self.oms[0], self.E, stream_ptr()),
              "natinf_flow_input_f16")
        return out

    def step(self, k: int, x: torch.Tensor, v_text: torch.Tensor, v_null: torch.Tensor, noises: torch.Tensor,
             want_next: bool = True):
        x_next = self._x[k & 1]
        if x_next.data_ptr() == x.data_ptr():
            x_next = self._x[(k + 1) & 1]
        idx, val, n = self.rows.ptrs(k)
        r = self.rows.rows[k]
        flags = _lib.SD3_CFG_ON_VELOCITY if self.euler else 0
        check(lib.natinf_step_f16chain(ptr(x), ptr(v_text), ptr(v_null), ptr(noises), ptr(self.hist), ptr(self.mean),
                                       ptr(x_next) if want_next else None, idx, val, n, r.diag, self.totals[k], k,
                                       self.sig[k], self.sig[k + 1], self.oms[k + 1], self.cfg, flags, self.E,
                                       stream_ptr()), "natinf_step_f16chain")
        return self.mean, (x_next if want_next else None)
