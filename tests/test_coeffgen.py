"""Coefficient generators: regression against the shipped matrices and the marginal-coefficient
invariant the reference prints (sum_j C[k,j] = alpha_{k+1}, ||B[k,:]|| = sigma_{k+1}; src/Utils.py:12-27)."""
import numpy as np
import pytest
import torch

from naturaldiffusion_amd import coeffgen as G
from oracle import ni_oracle as O


@pytest.mark.parametrize("n", [18, 24])
def test_ddim_discrete_reproduces_shipped_matrices(repo_root, n):
    C, B, node = O.load_coeff_npz(repo_root / f"results/ddim/ddim_{n:03d}.npz")
    c, b, nd = G.ddim_discrete(n)
    assert c.shape == C.shape and b.shape == B.shape and nd.shape == node.shape
    assert np.abs(c - C).max() <= 4e-16 * np.abs(C).max() * n          # same closed form, different product order
    assert np.abs(b - B).max() <= 4e-16 * n
    assert np.allclose(nd, node, rtol=0, atol=1e-15)


def test_vp_continuous_ddim_marginals_and_equivalence(tmp_path):
    ts = G.quadratic_time_grid(18)
    C, B, node = G.ddim_vp_continuous(ts)
    # marginals: the start state is pure noise although alpha(t=1) = 0.0066 != 0, so the signal row sums fall short
    # of alpha_{k+1} by exactly alpha_0*sigma_{k+1}/sigma_0 and the noise coefficient is sigma_{k+1}/sigma_0
    a0, s0 = node[0, 1], node[0, 2]
    assert np.allclose(C.sum(axis=1), node[1:, 1] - a0 * node[1:, 2] / s0, atol=1e-12)
    assert np.allclose(B[:, 0], node[1:, 2] / s0, atol=1e-12)
    assert np.abs(C.sum(axis=1) - node[1:, 1]).max() < 7e-3
    assert np.allclose(np.triu(C, 1), 0)
    # NI with this matrix == the classical DDIM loop x <- (sigma_t/sigma_s) x + (alpha_t - alpha_s sigma_t/sigma_s) x0
    model = O.analytic_vp_model()
    g = torch.Generator().manual_seed(0)
    noise = torch.randn(2, 3, 32, 32, generator=g)
    stds = [float(s) for s in node[:-1, 2].astype(np.float32)]
    xs = O.cifar_ni_trajectory(model, noise, C, B, node, stds=stds)
    x = noise.double()
    for k in range(18):
        x0 = O.cifar_data_fn(model, x.float(), node[k, 0], node[k, 1], node[k, 2], std=stds[k])
        a = node[k + 1, 2] / node[k, 2]
        x = a * x + (node[k + 1, 1] - node[k, 1] * a) * x0
    assert (xs[-1].double() - x).abs().max() <= 2e-5 * x.abs().max()     # fp32 state in the NI loop vs fp64 here
    p = tmp_path / "ddim_vp_018.npz"
    G.save_coeff_matrix(p, C, B, node)
    C2, B2, n2 = O.load_coeff_npz(p)
    assert np.array_equal(C2, C) and np.array_equal(B2, B) and np.array_equal(n2, node)


def test_quadratic_grid_matches_shipped_weights(repo_root):
    _, _, node = O.load_coeff_npz(repo_root / "weights/step_15_weight_173.npz")
    assert np.allclose(G.quadratic_time_grid(15), node[:, 0], atol=1e-12)
    al, sg = G.vp_alpha_sigma(node[:, 0])
    assert np.allclose(al, node[:, 1], atol=1e-6) and np.allclose(sg, node[:, 2], atol=1e-6)


def _close(repo_root, rel, got, tol=1e-12):
    C, B, node = O.load_coeff_npz(repo_root / rel)
    c, b, nd = got
    assert c.shape == C.shape and b.shape == B.shape and nd.shape == node.shape, rel
    assert np.abs(c - C).max() <= tol * max(1.0, np.abs(C).max()), (rel, np.abs(c - C).max())
    assert np.abs(b - B).max() <= tol * max(1.0, np.abs(B).max()), (rel, np.abs(b - B).max())
    assert np.abs(nd - node).max() <= tol * max(1.0, np.abs(node).max()), (rel, np.abs(nd - node).max())


@pytest.mark.parametrize("n", [18, 24])
def test_tracer_generators_reproduce_shipped_matrices(repo_root, n):
    """K8 of SURVEY section 8c: every sampler family the reference ships a matrix for (the reference derives them with
    sympy; the float64 tracer agrees to rounding)."""
    _close(repo_root, f"results/dpmsolver/dpmsolver2s_{n:03d}.npz", G.dpmsolver_singlestep(n // 2, 2))
    _close(repo_root, f"results/dpmsolver/dpmsolver3s_{n:03d}.npz", G.dpmsolver_singlestep(n // 3, 3))
    _close(repo_root, f"results/dpmsolverpp/dpmsolverpp2s_{n:03d}.npz", G.dpmsolver_singlestep(n // 2, 2, data_prediction=True))
    _close(repo_root, f"results/dpmsolverpp/dpmsolverpp3s_{n:03d}.npz", G.dpmsolver_singlestep(n // 3, 3, data_prediction=True))
    _close(repo_root, f"results/euler_heun/ode_euler_{n:03d}.npz", G.vp_euler(n))
    _close(repo_root, f"results/euler_heun/sde_euler_{n:03d}.npz", G.vp_euler(n, stochastic=True))
    _close(repo_root, f"results/euler_heun/ode_heun_{n:03d}.npz", G.vp_heun(n // 2))
    _close(repo_root, f"results/flow_euler/flow_euler_simpy_{n:03d}.npz", G.flow_euler(n))
    _close(repo_root, f"results/ddpm/ddpm_sympy_{n:03d}.npz", G.ddpm_discrete(n))


def test_marginal_invariants_of_generated_matrices():
    """what the reference prints as its own check (src/Utils.py:14-27): row sums of C follow alpha, row norms of B
    follow sigma -- exactly for DDPM / flow, to discretisation error for the ODE / SDE solvers."""
    C, B, node = G.ddpm_discrete(50)
    # the chain starts from pure noise although sqrt(abar_999) = 0.0064: the signal sums fall short by that much at most
    assert np.allclose(C.sum(1), node[1:, 1], atol=7e-3) and np.allclose(np.linalg.norm(B, axis=1), node[1:, 2], atol=7e-3)
    C, B, node = G.flow_euler(40)
    assert np.allclose(C.sum(1), node[1:, 1], atol=1e-12) and np.allclose(np.abs(B).sum(1), node[1:, 2], atol=1e-12)
    for C, B, node in (G.dpmsolver_singlestep(10, 3), G.dpmsolver_singlestep(12, 2, data_prediction=True), G.vp_heun(20, reference_quirks=False)):
        assert np.allclose(np.triu(C, 1), 0)
        assert np.abs(C.sum(1)[-1] - node[-1, 1]) < 2e-2


def test_published_dpmsolverpp3s_differs_from_the_reference_variant_and_is_more_accurate():
    """the sign the reference uses in DPM-Solver++(3S) (AnalyzeDPMSolver.py:597-613) against the published one, on a model
    whose exact solution is known: x0_hat = const makes the data-prediction ODE solution x_t = (sigma_t/sigma_s) x_s +
    alpha_t (1 - e^{-h}) x0 for any solver order, so both variants must be exact -- the difference shows on a
    t-dependent prediction."""
    Cq, _, node = G.dpmsolver_singlestep(6, 3, data_prediction=True, reference_sign=True)
    Cp, _, _ = G.dpmsolver_singlestep(6, 3, data_prediction=True, reference_sign=False)
    assert np.abs(Cq - Cp).max() > 1e-3
    assert np.allclose(Cq.sum(1), Cp.sum(1), atol=1e-12)        # constant prediction: identical (difference terms vanish)


def test_deis_tab_reproduces_shipped_matrix(repo_root):
    """tAB-DEIS: th_deis integrates in jax float32, so agreement with the shipped file is ~1e-5, and exact on the shipped
    3-decimal CSVs of the step counts that have no .npz."""
    import pandas as pd
    _close(repo_root, "results/deis/deis_tab_100.npz", G.deis_tab(100), tol=2e-5)
    for n in (18, 24):
        df = pd.read_csv(repo_root / f"results/deis/deis_tab_{n:03d}.csv", index_col=0)
        C, _, _ = G.deis_tab(n)
        assert np.array_equal(C.round(3), df.to_numpy()[:, :n])
        assert np.allclose(C.sum(1).round(3), df["sum"].to_numpy(), atol=1.1e-3)
