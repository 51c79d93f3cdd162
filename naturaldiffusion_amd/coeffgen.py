"""Coefficient matrices of first-order deterministic samplers in Natural Inference form (host-side numpy).

A sampler whose step is linear in the current state and the predicted x0,

    x_{i+1} = a_i * x_i + b_i * y_i ,     y_i = x0_hat(x_i, t_i) ,     x_0 = eps ,

unrolls to  x_{k+1} = sum_{j<=k} C[k,j] y_j + B[k,0] eps  with

    C[k,j] = b_j * prod_{m=j+1..k} a_m ,        B[k,0] = prod_{m=0..k} a_m .

This is the closed form behind the reference's ``ddim_analyze_coeff`` (src/AnalyzeDDPMDDIM.py:297-340).  Provided
here (SURVEY.md section 8f N1, and the DDIM matrix BASELINE config 3 needs on the VP-continuous grid):

* :func:`ddim_discrete`       -- DDIM on the 1000-step linear-beta DDPM schedule, strided (reproduces the shipped
                                 ``results/ddim/ddim_0NN.npz``),
* :func:`ddim_vp_continuous`  -- DDIM == DPM-Solver-1 on a continuous VP time grid (``a = sigma_t/sigma_s``,
                                 ``b = alpha_t - alpha_s*sigma_t/sigma_s``, cf. deps/dpm_solver_pytorch.py:547-592),
* :func:`save_coeff_matrix`   -- the reference's ``.npz`` layout (src/Utils.py:49).
Files written here load through the same positional reader as the shipped ones."""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np


def first_order_matrices(a: Sequence[float], b: Sequence[float]) -> Tuple[np.ndarray, np.ndarray]:
    """(C [N,N] lower-triangular, B [N,N+1] with only column 0 used) for x_{i+1} = a_i x_i + b_i y_i."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    n = len(a)
    C, B = np.zeros((n, n)), np.zeros((n, n + 1))
    for k in range(n):
        run = 1.0                                   # prod_{m=j+1..k} a_m, built from j = k downwards
        for j in range(k, -1, -1):
            C[k, j] = b[j] * run
            run *= a[j]
        B[k, 0] = run
    return C, B


def _strided_abar(num_step: int):
    betas = np.linspace(0.0001, 0.02, 1000, dtype=np.float64)
    abar = np.cumprod(1 - betas)
    stride = 999 / (num_step - 1) if num_step > 1 else 1
    idx = sorted({round(i * stride) for i in range(num_step)})
    return np.asarray(idx), abar[idx]


def ddim_discrete(num_step: int):
    """DDIM over ``num_step`` strided timesteps of the DDPM schedule -> (C, B, node_coeff) as shipped."""
    idx, ab = _strided_abar(num_step)
    prev = np.append(1.0, ab[:-1])
    rect = np.sqrt((1 - prev) / (1 - ab))                 # x_t coefficient of the reverse step at level i
    c_x0 = np.sqrt(prev) - rect * np.sqrt(ab)
    # sampling order runs from the noisiest level down: step s uses level num_step-1-s
    a, b = rect[::-1], c_x0[::-1]
    C, B = first_order_matrices(a, b)
    node = np.zeros((num_step + 1, 3))
    node[0] = (999, 0.0, 1.0)
    for k in range(num_step):                             # state after step k sits at level num_step-2-k
        lvl = num_step - 2 - k
        node[k + 1] = (idx[lvl], np.sqrt(ab[lvl]), np.sqrt(1 - ab[lvl])) if lvl >= 0 else (-1, 1.0, 0.0)
    return C, B, node


def vp_alpha_sigma(t, beta_0: float = 0.1, beta_1: float = 20.0):
    t = np.asarray(t, np.float64)
    lmc = -0.25 * t ** 2 * (beta_1 - beta_0) - 0.5 * t * beta_0
    return np.exp(lmc), np.sqrt(1.0 - np.exp(2.0 * lmc))


def quadratic_time_grid(num_step: int, t_start: float = 1.0, t_end: float = 1e-3) -> np.ndarray:
    """the ``time_quadratic`` grid of the shipped ``weights/step_*`` files: t_i = (sqrt(t0) + i/N (sqrt(tN)-sqrt(t0)))^2."""
    i = np.arange(num_step + 1, dtype=np.float64) / num_step
    return (np.sqrt(t_start) + i * (np.sqrt(t_end) - np.sqrt(t_start))) ** 2


def ddim_vp_continuous(ts: Sequence[float]):
    """DDIM / DPM-Solver-1 on the continuous VP SDE over the decreasing time grid ``ts`` (N+1 nodes)."""
    ts = np.asarray(ts, np.float64)
    al, sg = vp_alpha_sigma(ts)
    a = sg[1:] / sg[:-1]
    b = al[1:] - al[:-1] * a
    C, B = first_order_matrices(a, b)
    node = np.stack([ts, al, sg], axis=1)
    return C, B, node


def save_coeff_matrix(path, C, B, node) -> None:
    np.savez(path, past_xstart_coeff=np.asarray(C, np.float64), past_epsilon_coeff=np.asarray(B, np.float64),
             node_coeff=np.asarray(node, np.float64))
