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
* :func:`save_coeff_matrix`   -- the reference's ``.npz`` layout (src/Utils.py:49),
* :class:`LinearTrace` and, built on it, every other sampler family the reference ships matrices for:
  :func:`dpmsolver_singlestep` (DPM-Solver-2/-3, DPM-Solver++(2S)/(3S); src/AnalyzeDPMSolver.py),
  :func:`vp_euler` (ODE and SDE), :func:`vp_heun` (src/AnalyzeEulerHeun.py), :func:`flow_euler`
  (src/AnalyzeFlowMatching.py), :func:`ddpm_discrete` (src/AnalyzeDDPMDDIM.py) -- each regression-tested against the
  shipped ``results/**/*.npz`` to 1e-12 (tests/test_coeffgen.py); :func:`deis_tab` (tAB-DEIS, src/AnalyzeDEIS.py +
  deps/th_deis/multistep.py, without jax) to the float32 accuracy of the shipped file.
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


# --------------------------------------------------------------------------------------------------------------
# Generic route: any sampler whose update is linear in (state, predicted x0, injected noises) can be unrolled by
# carrying, instead of the state, its coefficient vector over the symbols  y_0, y_1, ... (predicted x0 of every model
# evaluation, in evaluation order) and  eps_0, eps_1, ... (initial noise, then every injected noise).  The reference
# does the same with sympy expressions and ``expr.coeff(symbol)`` (src/Utils.py:56-92); plain float64 vectors give the
# same matrices to rounding (regression-tested against the shipped files to 1e-12).
# --------------------------------------------------------------------------------------------------------------
class LinearTrace:
    """Book-keeping for one unrolled sampler with ``n_eval`` model evaluations."""

    def __init__(self, n_eval: int):
        self.n = int(n_eval)
        self._ny, self._ne = 0, 0
        self.nodes = []                                   # (t, alpha, sigma, state vector) in evaluation / time order

    def _vec(self):
        return np.zeros(2 * self.n + 1)

    def new_eps(self) -> np.ndarray:
        """a fresh unit-variance noise symbol (eps_0 is the initial state)"""
        v = self._vec()
        v[self.n + self._ne] = 1.0
        self._ne += 1
        return v

    def new_y(self) -> np.ndarray:
        """the predicted x0 of the next model evaluation"""
        v = self._vec()
        v[self._ny] = 1.0
        self._ny += 1
        return v

    def record(self, t, alpha, sigma, x) -> None:
        self.nodes.append((float(t), float(alpha), float(sigma), np.array(x, np.float64)))

    def matrices(self):
        """(C [n,n], B [n,n+1], node [n+1,3]); node 0 is the initial state, row k-1 is the state at node k."""
        assert len(self.nodes) == self.n + 1, (len(self.nodes), self.n)
        C, B, node = np.zeros((self.n, self.n)), np.zeros((self.n, self.n + 1)), np.zeros((self.n + 1, 3))
        for k, (t, al, sg, x) in enumerate(self.nodes):
            node[k] = (t, al, sg)
            if k:
                C[k - 1], B[k - 1] = x[:self.n], x[self.n:]
        return C, B, node


def vp_lambda(t, beta_0: float = 0.1, beta_1: float = 20.0):
    """half log-SNR  log(alpha_t) - log(sigma_t)  of the linear VP schedule"""
    t = np.asarray(t, np.float64)
    lmc = -0.25 * t ** 2 * (beta_1 - beta_0) - 0.5 * t * beta_0
    return lmc - 0.5 * np.log(1.0 - np.exp(2.0 * lmc))


def vp_inverse_lambda(lam, beta_0: float = 0.1, beta_1: float = 20.0):
    """t with vp_lambda(t) = lam (closed form for the linear schedule, Lu et al. 2022, appendix)"""
    tmp = 2.0 * (beta_1 - beta_0) * np.logaddexp(-2.0 * np.asarray(lam, np.float64), 0.0)
    return tmp / (np.sqrt(beta_0 ** 2 + tmp) + beta_0) / (beta_1 - beta_0)


def dpmsolver_singlestep(step: int, order: int, data_prediction: bool = False, reference_sign: bool = True,
                         t_start: float = 1.0, t_end: float = 1e-3):
    """Single-step DPM-Solver-2 / -3 (noise prediction) or DPM-Solver++ (2S) / (3S) (data prediction) over ``step`` outer
    steps on ``linspace(t_start, t_end, step+1)``: ``order*step`` model evaluations, intermediate nodes at
    ``lambda_s + r h`` with r = 1/2 (order 2) or 1/3, 2/3 (order 3).  Follows src/AnalyzeDPMSolver.py:228-667, i.e. the
    published update formulas with phi_1 = e^{+-h} - 1 and phi_2 = phi_1/h -+ 1.

    ``reference_sign``: the reference's DPM-Solver++(3S) subtracts the phi_2 difference terms
    (AnalyzeDPMSolver.py:597-613) where the published algorithm adds them; True reproduces the shipped
    ``results/dpmsolverpp/dpmsolverpp3s_*.npz``, False gives the published solver.  No effect on the other three."""
    if order not in (2, 3):
        raise ValueError("order must be 2 or 3")
    tr = LinearTrace(order * step)
    ts = np.linspace(t_start, t_end, step + 1)
    a0, s0 = vp_alpha_sigma(ts[0])
    x = tr.new_eps()
    tr.record(ts[0], a0, s0, x)
    rs = (0.5,) if order == 2 else (1.0 / 3.0, 2.0 / 3.0)
    sgn = -1.0 if data_prediction else 1.0                 # exponent sign: e^{h} (noise) / e^{-h} (data)
    for i in range(step):
        s, t = ts[i], ts[i + 1]
        lam_s, lam_t = vp_lambda(s), vp_lambda(t)
        h = lam_t - lam_s
        inter = [float(vp_inverse_lambda(lam_s + r * h)) for r in rs]
        al_s, sg_s = vp_alpha_sigma(s)

        def model(x_at, t_at):                             # what the update formulas call the model output
            y = tr.new_y()
            if data_prediction:
                return y
            al, sg = vp_alpha_sigma(t_at)
            return (x_at - al * y) / sg                     # predicted noise from predicted x0

        def base(t_to):                                     # first-order part towards t_to: carry term and its own scale
            al, sg = vp_alpha_sigma(t_to)
            return (sg / sg_s, al) if data_prediction else (al / al_s, sg)
        m_s = model(x, s)
        # stage 1
        r1 = rs[0]
        carry, scale = base(inter[0])
        x1 = carry * x - scale * np.expm1(sgn * r1 * h) * m_s
        al1, sg1 = vp_alpha_sigma(inter[0])
        tr.record(inter[0], al1, sg1, x1)
        m_1 = model(x1, inter[0])
        if order == 2:
            carry, scale = base(t)
            phi = np.expm1(sgn * h)
            xt = carry * x - scale * phi * m_s - (0.5 / r1) * scale * phi * (m_1 - m_s)
        else:
            r2 = rs[1]
            flip = -1.0 if (data_prediction and not reference_sign) else 1.0
            carry, scale = base(inter[1])
            phi1 = np.expm1(sgn * r2 * h)
            phi2 = phi1 / (r2 * h) - sgn
            x2 = carry * x - scale * phi1 * m_s - flip * (r2 / r1) * scale * phi2 * (m_1 - m_s)
            al2, sg2 = vp_alpha_sigma(inter[1])
            tr.record(inter[1], al2, sg2, x2)
            m_2 = model(x2, inter[1])
            carry, scale = base(t)
            phi1 = np.expm1(sgn * h)
            phi2 = phi1 / h - sgn
            xt = carry * x - scale * phi1 * m_s - flip * (1.0 / r2) * scale * phi2 * (m_2 - m_s)
        al_t, sg_t = vp_alpha_sigma(t)
        tr.record(t, al_t, sg_t, xt)
        x = xt
    return tr.matrices()


def _vp_drift_diffusion(t, beta_0: float = 0.1, beta_1: float = 20.0):
    beta = beta_0 + t * (beta_1 - beta_0)
    return -0.5 * beta, np.sqrt(beta)


def _uniform_vp_grid(num_step: int) -> np.ndarray:
    """t_i = 1 + i (eta - 1)/(N - 1), i = 0..N-1, with N = num_step + 1 and eta = 1/N (AnalyzeEulerHeun.py:52-57)"""
    N = num_step + 1
    return 1.0 + np.arange(N) * (1.0 / N - 1.0) / (N - 1)


def vp_euler(num_step: int, stochastic: bool = False):
    """Euler on the VP probability-flow ODE, or Euler-Maruyama on the reverse SDE (``stochastic``), with the score
    written through the predicted x0: score = (alpha y - x)/sigma^2 (src/AnalyzeEulerHeun.py:50-123, 125-200)."""
    tr = LinearTrace(num_step)
    ts = _uniform_vp_grid(num_step)
    dt = ts[1] - ts[0]
    x = tr.new_eps()
    tr.record(ts[0], *vp_alpha_sigma(ts[0]), x)
    for i in range(num_step):
        s = ts[i]
        al, sg = vp_alpha_sigma(s)
        f, g = _vp_drift_diffusion(s)
        score = (al * tr.new_y() - x) / sg ** 2
        if stochastic:
            x = x + (f * x - g ** 2 * score) * dt + g * np.sqrt(abs(dt)) * tr.new_eps()
        else:
            x = x + (f * x - 0.5 * g ** 2 * score) * dt
        tr.record(ts[i + 1], *vp_alpha_sigma(ts[i + 1]), x)
    return tr.matrices()


def vp_heun(num_step: int, reference_quirks: bool = True):
    """Heun (2nd order) on the VP probability-flow ODE: ``2*num_step`` evaluations, the predictor state is a node of its
    own (src/AnalyzeEulerHeun.py:203-292).

    ``reference_quirks`` (True reproduces the shipped ``ode_heun_*.npz``): the reference labels the predictor node
    ``t + 0.0005`` (and takes its (alpha, sigma) there), and forms the corrector's score with alpha_s instead of alpha_t
    (``:243``).  False uses time t and alpha_t."""
    tr = LinearTrace(2 * num_step)
    ts = _uniform_vp_grid(num_step)
    dt = ts[1] - ts[0]
    x = tr.new_eps()
    tr.record(ts[0], *vp_alpha_sigma(ts[0]), x)
    for i in range(num_step):
        s, t = ts[i], ts[i + 1]
        al_s, sg_s = vp_alpha_sigma(s)
        al_t, sg_t = vp_alpha_sigma(t)
        f_s, g_s = _vp_drift_diffusion(s)
        f_t, g_t = _vp_drift_diffusion(t)
        v_s = f_s * x - 0.5 * g_s ** 2 * (al_s * tr.new_y() - x) / sg_s ** 2
        xh = x + v_s * dt
        th = t + 0.0005 if reference_quirks else t
        tr.record(th, *vp_alpha_sigma(th), xh)
        v_t = f_t * xh - 0.5 * g_t ** 2 * ((al_s if reference_quirks else al_t) * tr.new_y() - xh) / sg_t ** 2
        x = x + 0.5 * (v_s + v_t) * dt
        tr.record(t, al_t, sg_t, x)
    return tr.matrices()


def flow_euler(num_step: int):
    """Euler on the rectified-flow ODE x_t = (1-t) x0 + t eps over t = 1 -> 0 in ``num_step`` uniform steps, velocity
    (x - y)/t (src/AnalyzeFlowMatching.py:62-115)."""
    tr = LinearTrace(num_step)
    ts = np.linspace(0.0, 1.0, num_step + 1)[::-1]
    x = tr.new_eps()
    tr.record(ts[0], 1.0 - ts[0], ts[0], x)
    for i in range(num_step):
        s, t = ts[i], ts[i + 1]
        x = x + (x - tr.new_y()) / s * (t - s)
        tr.record(t, 1.0 - t, t, x)
    return tr.matrices()


def space_timesteps_uniform(num_timesteps: int, count: int):
    """the single-section case of ``space_timesteps(num_timesteps, str(count))`` (improved-DDPM; reference
    src/AnalyzeDDPMDDIM.py:23-73): ``count`` indices from 0 to num_timesteps-1 at a fractional stride, rounded."""
    if count > num_timesteps:
        raise ValueError("cannot divide section of %d steps into %d" % (num_timesteps, count))
    stride = 1 if count <= 1 else (num_timesteps - 1) / (count - 1)
    return sorted({round(stride * i) for i in range(count)})


def ddpm_discrete(num_step: int):
    """Ancestral DDPM sampling over ``num_step`` strided timesteps of the 1000-step linear-beta schedule:
    x_prev = c_xt x + c_x0 y + std eps (posterior mean / variance of the strided chain, first log-variance clipped to
    log 1e-5; src/AnalyzeDDPMDDIM.py:76-123,177-247).  Reproduces ``results/ddpm/ddpm_sympy_*.npz``."""
    betas = np.linspace(0.0001, 0.02, 1000, dtype=np.float64)
    abar_all = np.cumprod(1.0 - betas)
    idx = space_timesteps_uniform(1000, num_step)
    abar = abar_all[idx]
    alpha = np.append(abar[0], abar[1:] / abar[:-1])
    beta = 1.0 - alpha
    prev = np.append(1.0, abar[:-1])
    var = beta * (1.0 - prev) / (1.0 - abar)
    std = np.sqrt(np.exp(np.log(np.append(1e-5, var[1:]))))
    c_x0 = np.sqrt(prev) * beta / (1.0 - abar)
    c_xt = np.sqrt(alpha) * (1.0 - prev) / (1.0 - abar)
    tr = LinearTrace(num_step)
    x = tr.new_eps()
    # the reference lists the start node as (t, sqrt(abar_t), sqrt(1 - sqrt(abar_t)^2)) (:236-239)
    top = np.sqrt(abar[-1])
    tr.record(idx[-1], top, np.sqrt(1.0 - top ** 2), x)
    for k in range(num_step):
        lvl = num_step - 1 - k
        x = c_xt[lvl] * x + c_x0[lvl] * tr.new_y() + std[lvl] * tr.new_eps()
        a = np.sqrt(abar[lvl - 1]) if lvl > 0 else 1.0
        tr.record(idx[lvl - 1] if lvl > 0 else -1, a, np.sqrt(1.0 - a ** 2), x)
    return tr.matrices()


def deis_tab(num_step: int, order: int = 3, t_end: float = 1e-3, t_start: float = 1.0, n_quad: int = 10000,
             beta_0: float = 0.1, beta_1: float = 20.0):
    """tAB-DEIS (Zhang & Chen 2022; src/AnalyzeDEIS.py with deps/th_deis/multistep.py:6-97): exponential integrator with the
    noise prediction extrapolated by the Lagrange polynomial through the last ``order+1`` evaluations,

        x_{i+1} = Psi(t_i, t_{i+1}) x_i + sum_j C_ij eps_hat_{i-j} ,
        C_ij = int_{t_i}^{t_{i+1}} Psi(tau, t_{i+1}) * (-1/2 dlog(abar)/dtau / sqrt(1 - abar(tau))) * L_j(tau) dtau ,

    on the quadratic time grid t_i = (sqrt(T) + i/N (sqrt(eps) - sqrt(T)))^2, order ramping 0, 1, .. over the first steps.
    The integral is taken as th_deis takes it -- a left Riemann sum over ``n_quad`` points -- but in float64 (th_deis runs
    jax's float32), so the shipped ``results/deis/deis_tab_*.npz`` are reproduced to ~1e-5, not to rounding."""
    def log_abar(t):
        return -0.5 * t ** 2 * (beta_1 - beta_0) - t * beta_0
    ts = np.linspace(np.sqrt(t_start), np.sqrt(t_end), num_step + 1) ** 2
    tr = LinearTrace(num_step)
    x = tr.new_eps()
    tr.record(ts[0], *vp_alpha_sigma(ts[0], beta_0, beta_1), x)
    hist = []                                               # predicted noises, newest first
    for i in range(num_step):
        t0, t1 = ts[i], ts[i + 1]
        al, sg = vp_alpha_sigma(t0, beta_0, beta_1)
        hist.insert(0, (x - al * tr.new_y()) / sg)
        o = min(i, order)
        nodes = ts[i - o:i + 1]
        tau = np.linspace(t0, t1, n_quad, endpoint=False)
        dt = (t1 - t0) / n_quad
        w = np.exp(0.5 * (log_abar(t1) - log_abar(tau))) * (0.5 * (tau * (beta_1 - beta_0) + beta_0) / np.sqrt(1.0 - np.exp(log_abar(tau))))
        xn = np.exp(0.5 * (log_abar(t1) - log_abar(t0))) * x
        for j in range(o + 1):                              # j = 0: the newest evaluation (node t_i)
            idx = o - j
            L = np.ones_like(tau)
            for k in range(o + 1):
                if k != idx:
                    L *= (tau - nodes[k]) / (nodes[idx] - nodes[k])
            xn = xn + np.sum(w * L) * dt * hist[j]
        x = xn
        hist = hist[:order]
        tr.record(t1, *vp_alpha_sigma(t1, beta_0, beta_1), x)
    return tr.matrices()
