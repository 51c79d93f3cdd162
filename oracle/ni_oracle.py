"""Oracle: the Natural Inference recurrence, three arithmetic flavours.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  torch-CPU restatement of

* CIFAR10 form  -- ``src/CIFAR10NaturalInference.py:219-238,287-309`` (fp64
  history, fp64 accumulate, fp32 model I/O), score wrapper
  ``deps/score_sde_pytorch/models/utils.py:144-160`` and VP schedule
  ``deps/score_sde_pytorch/sde_lib.py:141-145``;
* Validate form -- ``src/ValidateNaturalInference.py:177-204,311-372`` (fp32
  products accumulated in fp64, per-step fresh noise, CFG fuse);
* SD3 form      -- ``src/SD3NaturalInference.py:61-69,157-168,198-223`` (all-fp16
  chain, row-normalised weighted mean, CFG 7) and its Euler twin ``:81-154``.

The restatement is explicit about every rounding: each function spells out the
dtype in which a product / sum is formed, mirroring what eager PyTorch does for
"tensor (op) python-or-numpy scalar" (the scalar is converted to the tensor's
op-math type: fp32 for fp16/fp32 tensors, fp64 for fp64 tensors) and for
"tensor (op) 0-d tensor" (the 0-d tensor is first cast to the result dtype).
"""
from __future__ import annotations

import csv
import math
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

F16, F32, F64 = torch.float16, torch.float32, torch.float64


# --------------------------------------------------------------------------- #
# coefficient files  (Utils.py:30-53 writer; CIFAR10...:273, SD3...:196 readers)
# --------------------------------------------------------------------------- #
def load_coeff_npz(path) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Positional ``.values()`` order (C, B, node_coeff) -- CIFAR10...:273."""
    with np.load(path) as z:
        vals = [z[k] for k in z.files]
    assert len(vals) == 3, "coefficient npz must hold exactly three arrays"
    C, B, node = (np.asarray(v, dtype=np.float64) for v in vals)
    return C, B, node


def load_sd3_csv(path) -> np.ndarray:
    """``pd.read_csv(path, index_col=0).to_numpy()`` -- SD3...:196, without pandas.

    First row = column labels, first column = row labels, body = weights.
    """
    with open(path, newline="") as fh:
        rows = list(csv.reader(fh))
    body = [[float(v) for v in r[1:]] for r in rows[1:] if len(r) > 1]
    return np.asarray(body, dtype=np.float64)


def sd3_sigma_schedule(num_step: int = 28, shift: float = 3.0, n_train: int = 1000):
    """FlowMatchEulerDiscreteScheduler.set_timesteps(num_step) of SD3-medium
    (shift 3): returns (timesteps fp32 [N], sigmas fp32 [N+1]).  Third-party
    arithmetic restated from its published definition; pinned by the diagonal of
    ``weights/sd3_step_28_weight.csv``."""
    s_all = np.linspace(1, n_train, n_train, dtype=np.float32)[::-1].copy() / n_train
    s_all = shift * s_all / (1 + (shift - 1) * s_all)
    s_max, s_min = float(s_all[0]), float(s_all[-1])
    ts = np.linspace(s_max * n_train, s_min * n_train, num_step)
    sig = ts / n_train
    sig = shift * sig / (1 + (shift - 1) * sig)
    sig32 = torch.from_numpy(sig).to(F32)
    timesteps = sig32 * n_train
    sigmas = torch.cat([sig32, torch.zeros(1, dtype=F32)])
    return timesteps, sigmas


# --------------------------------------------------------------------------- #
# CIFAR10 form
# --------------------------------------------------------------------------- #
def vp_std_f32(t: float, beta_0: float = 0.1, beta_1: float = 20.0) -> torch.Tensor:
    """std(t) of the VP SDE evaluated the way ``score_fn`` does it: on an fp32
    vector ``t * ones`` (sde_lib.py:141-145).  Returns a 0-d fp32 tensor."""
    vt = torch.ones(1, dtype=F32) * t              # np.float64 * fp32 tensor -> fp32
    lmc = -0.25 * vt ** 2 * (beta_1 - beta_0) - 0.5 * vt * beta_0
    return torch.sqrt(1.0 - torch.exp(2.0 * lmc))[0]


def score_from_model_out(out: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
    """models/utils.py:157 -- ``score = -out / std`` in fp32."""
    return (-out) / std


def x0_from_score(xt: torch.Tensor, score: torch.Tensor, alpha: float, sigma: float) -> torch.Tensor:
    """CIFAR10...:224-229: fp64 ``(score * sigma**2 + xt) / alpha``, separate roundings."""
    xt64, sc64 = xt.to(F64), score.to(F64)
    s = torch.tensor(float(sigma), dtype=F64)
    a = torch.tensor(float(alpha), dtype=F64)
    return (sc64 * (s * s) + xt64) / a


def cifar_data_fn(model_fn: Callable, xt: torch.Tensor, t: float, alpha: float, sigma: float,
                  std: Optional[float] = None) -> torch.Tensor:
    """A1+A2.  ``model_fn(x, labels)`` is the raw network (labels = t*999, fp32).  ``std`` overrides the fp32
    VP std (torch.exp is machine-dependent in the last ulp; fixtures carry the value they were made with)."""
    vec_t = torch.ones(xt.shape[0], dtype=F32) * t
    out = model_fn(xt, vec_t * 999)
    std = vp_std_f32(t) if std is None else torch.tensor(std, dtype=F32)
    return x0_from_score(xt, score_from_model_out(out, std), alpha, sigma)


def cifar_weighted_sum(coeff_row: Sequence[float], seq_x0: Sequence[torch.Tensor]) -> torch.Tensor:
    """A3 (CIFAR10...:233-238): fp64 mul, fp64 add, ascending order, then fp32."""
    acc = torch.zeros_like(seq_x0[0], dtype=F64)
    for j, h in enumerate(seq_x0):
        acc = acc + h * float(coeff_row[j])
    return acc.to(F32)


def cifar_ni_trajectory(model_fn: Callable, noise: torch.Tensor, C: np.ndarray, B: np.ndarray,
                        node: np.ndarray, stds: Optional[Sequence[float]] = None) -> List[torch.Tensor]:
    """A5 (CIFAR10...:292-304).  Returns [x_0 (=noise), x_1, ..., x_N] (fp32)."""
    n_step = node.shape[0] - 1
    xs, hist = [noise], []
    x = noise
    for k in range(n_step):
        hist.append(cifar_data_fn(model_fn, x, node[k, 0], node[k, 1], node[k, 2],
                                  None if stds is None else float(stds[k])))
        nxt = cifar_weighted_sum(C[k], hist)
        eps = noise * float(np.float32(B[k, 0]))                    # fp32 scalar * fp32 tensor
        x = nxt + eps
        xs.append(x)
    return xs


def to_pixel(x: torch.Tensor) -> torch.Tensor:
    """inverse scaler (datasets.py:32-38) + to_pixel (CIFAR10...:212-216): uint8 NHWC."""
    y = ((x + 1.0) / 2.0).permute(0, 2, 3, 1).numpy()
    return torch.from_numpy(np.clip(y * 255, 0, 255).astype(np.uint8))


# --------------------------------------------------------------------------- #
# Validate (DiT, epsilon-prediction) form
# --------------------------------------------------------------------------- #
def validate_weighted_sum(weights: Sequence[float], seq: Sequence[torch.Tensor]) -> torch.Tensor:
    """A6 (Validate...:198-204): fp32 product (scalar -> fp32), fp64 accumulate, -> fp32."""
    acc = torch.zeros_like(seq[0], dtype=F64)
    for j, e in enumerate(seq):
        prod = e * float(np.float32(weights[j]))                     # fp32
        acc = acc + prod.to(F64)
    return acc.to(F32)


def cfg_fuse(cond: torch.Tensor, uncond: torch.Tensor, scale: float) -> torch.Tensor:
    """Validate...:193 -- ``uncond + scale*(cond-uncond)``, every op rounded in the tensor dtype."""
    return uncond + (cond - uncond) * scale


def ddim_skip_tables(num_step: int):
    """``skip_ddim_coeff(create_ddim_coeff(), n)`` (Validate...:136-174), numpy fp64."""
    betas = np.linspace(1e-4, 2e-2, 1000, dtype=np.float64)
    abar = np.cumprod(1.0 - betas)
    idx = sorted(_space_timesteps(1000, num_step))
    sab = abar[idx]
    sab_prev = np.append(1.0, sab[:-1])
    xt2x0 = np.sqrt(1.0 / sab)
    eps2x0 = np.sqrt(1.0 / sab - 1)
    rect = np.sqrt((1 - sab_prev) / (1 - sab))
    c_x0 = np.sqrt(sab_prev) - rect * np.sqrt(sab)
    return dict(idx=idx, abar=sab, xt2x0=xt2x0, eps2x0=eps2x0, c_xt=rect, c_x0=c_x0)


def ddpm_skip_tables(num_step: int):
    """``skip_ddpm_coeff(create_ddpm_coeff(), n)`` (Validate...:82-133), numpy fp64."""
    betas = np.linspace(1e-4, 2e-2, 1000, dtype=np.float64)
    abar = np.cumprod(1.0 - betas)
    idx = sorted(_space_timesteps(1000, num_step))
    sab = abar[idx]
    sa = np.empty_like(sab)
    sa[0] = sab[0]
    sa[1:] = sab[1:] / sab[:-1]
    sb = 1 - sa
    sab_prev = np.append(1.0, sab[:-1])
    var = sb * (1.0 - sab_prev) / (1.0 - sab)
    log_var = np.log(np.append(1e-5, var[1:]))
    c_x0 = np.sqrt(sab_prev) * sb / (1 - sab)
    c_xt = np.sqrt(sa) * (1 - sab_prev) / (1 - sab)
    return dict(idx=idx, abar=sab, log_var=log_var, xt2x0=np.sqrt(1.0 / sab),
                eps2x0=np.sqrt(1.0 / sab - 1), c_xt=c_xt, c_x0=c_x0)


def _space_timesteps(n_total: int, count: int):
    """single-section case of ``space_timesteps`` (Validate...:28-79)."""
    if count <= 1:
        stride = 1.0
    else:
        stride = (n_total - 1) / (count - 1)
    cur, out = 0.0, []
    for _ in range(count):
        out.append(round(cur))
        cur += stride
    return set(out)


def validate_original(eps_fn: Callable, z0: torch.Tensor, noises: Sequence[torch.Tensor], num_step: int,
                      stochastic: bool) -> torch.Tensor:
    """``ddpm_skip_sample`` / ``ddim_skip_sample`` (Validate...:207-308) with the
    denoiser abstracted as ``eps_fn(z, int_timestep) -> fused eps`` and the RNG
    draws supplied (``noises[i]`` = the i-th ``randn_like``)."""
    tb = ddpm_skip_tables(num_step) if stochastic else ddim_skip_tables(num_step)
    f = lambda a: torch.from_numpy(a).to(F32)
    xt2x0, eps2x0, c_xt, c_x0 = f(tb["xt2x0"]), f(tb["eps2x0"]), f(tb["c_xt"]), f(tb["c_x0"])
    log_var = f(tb["log_var"]) if stochastic else None
    z, draw = z0, 0
    for i in reversed(range(num_step)):
        eps = eps_fn(z, tb["idx"][i])
        x0 = xt2x0[i] * z - eps2x0[i] * eps
        mean = c_xt[i] * z + c_x0[i] * x0
        if stochastic:
            z = mean + torch.exp(0.5 * log_var[i]) * noises[draw]
            draw += 1
        else:
            z = mean
    return z


def validate_ni(eps_fn: Callable, noise0: torch.Tensor, noises: Sequence[torch.Tensor], C: np.ndarray,
                B: np.ndarray, node: np.ndarray, return_all: bool = False):
    """A7 (Validate...:311-372): ``eps_fn(z, int(node[k,0]))`` returns the fused eps."""
    n_step = B.shape[0]
    tb = ddim_skip_tables(n_step)
    c1 = torch.from_numpy(tb["xt2x0"]).to(F32).flip(0)
    c2 = torch.from_numpy(tb["eps2x0"]).to(F32).flip(0)
    seq_x0, seq_eps, z, zs = [], [noise0], noise0, [noise0]
    for k in range(n_step):
        eps = eps_fn(z, int(node[k, 0]))
        seq_x0.append(c1[k] * z - c2[k] * eps)
        seq_eps.append(noises[k])
        z = validate_weighted_sum(C[k], seq_x0) + validate_weighted_sum(B[k], seq_eps)
        zs.append(z)
    return zs if return_all else z


# --------------------------------------------------------------------------- #
# SD3 (flow, fp16 chain) form
# --------------------------------------------------------------------------- #
def sd3_weighted_mean(seq: Sequence[torch.Tensor], weights: Optional[np.ndarray]) -> torch.Tensor:
    """A8 (SD3...:157-168): row ``len(seq)-1`` of ``weights``; product, running sum
    and the final divide are each rounded to the history dtype (fp16)."""
    n = len(seq)
    acc = torch.zeros_like(seq[0])
    tot = 0
    for j, h in enumerate(seq):
        w = 1 if weights is None else weights[n - 1][j]
        acc = acc + h * float(w)
        tot = tot + w
    return acc / float(tot)


def sd3_ni(velocity_fn: Callable, noises: torch.Tensor, weights: np.ndarray, sigmas: torch.Tensor,
           timesteps: torch.Tensor, cfg: float = 7.0, return_all: bool = False):
    """A9 (SD3...:198-223).  ``velocity_fn(x, t, cond: bool)`` -> fp16 velocity."""
    n_step = weights.shape[0]
    seq, means, mean = [], [], torch.zeros_like(noises)
    for k in range(n_step):
        sig = sigmas[k]                                    # 0-d fp32 tensor
        x = sig * noises + (1 - sig) * mean                # 0-d operands are cast to fp16 first
        v_text = velocity_fn(x, timesteps[k], True)
        v_null = velocity_fn(x, timesteps[k], False)
        x0_null = x - sig * v_null
        x0_text = x - sig * v_text
        seq.append(x0_null + cfg * (x0_text - x0_null))
        mean = sd3_weighted_mean(seq, weights)
        means.append(mean)
    return means if return_all else mean


def sd3_euler_ni(velocity_fn: Callable, noises: torch.Tensor, sigmas: torch.Tensor, timesteps: torch.Tensor,
                 cfg: float = 7.0, vanilla: bool = False) -> torch.Tensor:
    """A10 (SD3...:61-69, 81-154): flow-Euler, classical (``vanilla``) or NI update."""
    x, seq = noises.clone(), []
    out = None
    for i in range(len(timesteps)):
        v_text = velocity_fn(x, timesteps[i], True)
        v_null = velocity_fn(x, timesteps[i], False)
        v = v_null + cfg * (v_text - v_null)
        x0 = x - sigmas[i] * v
        seq.append((-1 * (sigmas[i + 1] - sigmas[i]), x0))
        acc, tot = torch.zeros_like(x0), 0
        for w, h in seq:
            acc = acc + w * h
            tot = tot + w
        out = acc / tot
        if vanilla:
            x = x + (sigmas[i + 1] - sigmas[i]) * v
        else:
            x = sigmas[i + 1] * noises + (1 - sigmas[i + 1]) * out
    return out


# --------------------------------------------------------------------------- #
# stand-in denoisers used by the fixtures
#
# Built ONLY from +, -, *, / on fp32 CPU tensors: those are correctly rounded by IEEE-754 on every machine, so
# the build container (where the fixtures were captured) and the GPU box feed ni_step the same bits.
# torch.exp / sin / cos / sqrt are NOT used: their vectorised CPU implementations differ in the last ulp
# between hosts (observed: the fp32 VP std of step 0 is 0.99997836 on one host and 0.99997842 on another, and
# torch.sqrt of an 8-element fp32 tensor differed between two AVX512 hosts).
# --------------------------------------------------------------------------- #
def _bump(x: torch.Tensor) -> torch.Tensor:
    """smooth bounded non-linearity x / (1 + x^2)."""
    return x / (1.0 + x * x)


def analytic_vp_model(mu: float = 0.25, s: float = 0.5, wobble: float = 0.05) -> Callable:
    """eps-hat of N(mu, s^2) data under the variance-preserving schedule a(t) = (1-t^2)/(1+t^2),
    sg(t) = 2t/(1+t^2) (a^2 + sg^2 = 1 without a square root), plus a small non-linear term.  ``labels = t*999``."""
    def model_fn(x: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        xc = x.detach().to("cpu", F32)
        t = labels.detach().to("cpu", F32) / 999
        a = ((1.0 - t * t) / (1.0 + t * t))[:, None, None, None]
        sg = ((2.0 * t) / (1.0 + t * t))[:, None, None, None]
        out = sg * (xc - a * mu) / (a * a * (s * s) + sg * sg) + wobble * _bump(xc)
        return out.to(x.device)
    return model_fn


def analytic_eps_model(wobble: float = 0.1) -> Callable:
    """fused-eps stand-in for the DiT+CFG call of the Validate form (fp32, CPU)."""
    abar_tab = np.cumprod(1.0 - np.linspace(1e-4, 2e-2, 1000))

    def eps_fn(z: torch.Tensor, timestep: int) -> torch.Tensor:
        zc = z.detach().to("cpu", F32)
        abar = float(abar_tab[max(int(timestep), 0)])
        out = zc * float(np.float32(1 - abar)) + wobble * _bump(3.0 * zc)
        return out.to(z.device)
    return eps_fn


def analytic_velocity_model() -> Callable:
    """velocity stand-in for the two MMDiT calls of the SD3 form (fp16 in/out, fp32 arithmetic)."""
    def vel_fn(x: torch.Tensor, t: torch.Tensor, cond: bool) -> torch.Tensor:
        xc = x.detach().to("cpu", F32)
        tt = float(t) / 1000.0
        tgt = 0.3 if cond else -0.1
        v = (xc - tgt) * (0.5 + 0.5 * tt) + 0.05 * _bump(2.0 * xc + (1.0 if cond else 0.0))
        return v.to(x.dtype).to(x.device)
    return vel_fn
