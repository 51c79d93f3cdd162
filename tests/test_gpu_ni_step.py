"""GPU parity tests of the ni_step kernels (through the C ABI) against the oracle and the golden
vectors captured from the reference.  Integer-free but bit-exact: the kernels reproduce the reference's
operand types, operation order and roundings, so every comparison below is ``array_equal``."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ni_oracle as O


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from naturaldiffusion_amd import _lib
    _lib.require_gpu()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def cifar(golden_dir):
    return np.load(golden_dir / "cifar_form.npz")


@pytest.fixture(scope="module")
def validate(golden_dir):
    return np.load(golden_dir / "validate_form.npz")


@pytest.fixture(scope="module")
def sd3(golden_dir):
    return np.load(golden_dir / "sd3_form.npz")


# ------------------------------------------------------------------------------ CIFAR10 form
@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("name", ["step_5_weight_00", "step_10_weight_42", "step_15_weight_173"])
def test_cifar_trajectory_bit_exact(dev, cifar, repo_root, name, dense):
    from naturaldiffusion_amd.CIFAR10NaturalInference import natural_inference, to_pixel_from_centered
    ref = cifar[f"k4_{name}_xs"]
    noise = torch.from_numpy(ref[0]).to(dev)
    xs = natural_inference(O.analytic_vp_model(), noise, repo_root / f"weights/{name}.npz", dense=dense, return_all=True,
                           stds=cifar[f"k4_{name}_stds"])     # host-dependent fp32 std pinned to the fixture's
    assert len(xs) == ref.shape[0]
    for k, x in enumerate(xs):
        assert np.array_equal(x.cpu().numpy(), ref[k]), f"x_{k} differs from the reference"
    assert np.array_equal(to_pixel_from_centered(xs[-1]).numpy(), cifar[f"k4_{name}_pix"])


@pytest.mark.parametrize("rel", ["dpmsolverpp/dpmsolverpp2s_018", "euler_heun/ode_euler_018"])
def test_classical_sampler_matrices(dev, cifar, repo_root, rel):
    from naturaldiffusion_amd.CIFAR10NaturalInference import natural_inference
    key = rel.split("/")[1]
    noise = torch.from_numpy(cifar[f"k5_{key}_noise"]).to(dev)
    out = natural_inference(O.analytic_vp_model(), noise, repo_root / f"results/{rel}.npz", stds=cifar[f"k5_{key}_stds"])
    assert np.array_equal(out.cpu().numpy(), cifar[f"k5_{key}_final"])


@pytest.mark.parametrize("rel", ["dpmsolverpp/dpmsolverpp2s_018", "dpmsolver/dpmsolver2s_018", "dpmsolver/dpmsolver3s_018"])
def test_ni_equals_the_vendored_classical_solver(dev, golden_dir, repo_root, rel):
    """SURVEY K5 on the HIP path: natinf_step_f64hist driven by the shipped DPM-Solver / DPM-Solver++ matrices against the
    output of the reference's own DPM_Solver class (deps/dpm_solver_pytorch.py:906, captured by make_golden.py group k5):
    bit-exact against the reference's NI loop, within the survey's 2e-5 of the classical solver."""
    from naturaldiffusion_amd.CIFAR10NaturalInference import natural_inference
    fx = np.load(golden_dir / "k5_classical.npz")
    key = rel.split("/")[1]
    noise = torch.from_numpy(fx[f"{key}_noise"]).to(dev)
    out = natural_inference(O.analytic_vp_model(), noise, repo_root / f"results/{rel}.npz", stds=fx[f"{key}_stds"]).cpu().numpy()
    assert np.array_equal(out, fx[f"{key}_ni"])
    assert np.abs(out - fx[f"{key}_orig"]).max() <= 2e-5


@pytest.mark.parametrize("key", ["lin18", "quad15"])
def test_ddim_on_the_vp_grid_through_the_hip_path(dev, golden_dir, tmp_path, key):
    """BASELINE config 3's DDIM leg on the HIP path: natinf_step_f64hist driven by the matrix coeffgen.ddim_vp_continuous writes in the
    reference's .npz format -- bit-exact against the reference's NI loop, within 2e-5 of the vendored DPM_Solver.dpm_solver_first_update
    (deps/dpm_solver_pytorch.py:547-592; tests/golden/make_golden.py group ddim_vp), dense and zero-skipped rows alike."""
    from naturaldiffusion_amd import coeffgen as G
    from naturaldiffusion_amd.CIFAR10NaturalInference import natural_inference
    fx = np.load(golden_dir / "ddim_vp.npz")
    C, B, node = G.ddim_vp_continuous(fx[f"{key}_ts"])
    path = tmp_path / f"ddim_vp_{key}.npz"
    G.save_coeff_matrix(path, C, B, node)
    noise = torch.from_numpy(fx[f"{key}_noise"]).to(dev)
    out = natural_inference(O.analytic_vp_model(), noise, path, stds=fx[f"{key}_stds"]).cpu().numpy()
    assert np.array_equal(out, fx[f"{key}_ni"])
    assert np.abs(out - fx[f"{key}_orig"]).max() <= 2e-5 and np.abs(out - fx[f"{key}_orig_pp"]).max() <= 2e-5


def test_data_fn_and_weighted_sum_mirrors(dev, cifar, repo_root):
    from naturaldiffusion_amd import CIFAR10NaturalInference as M
    C, B, node = O.load_coeff_npz(repo_root / "weights/step_15_weight_173.npz")
    xt = torch.from_numpy(cifar["k2_xt"]).to(dev)
    model = O.analytic_vp_model()

    for r in (0, 7, 14):
        def score_fn(x, vec_t, r=r):             # models/utils.py:144-160 with the stand-in network, fp32
            out = model(x, vec_t * 999).cpu()      # the division is done on the CPU (IEEE) so the score bits are the fixture's
            return ((-out) / torch.tensor(float(cifar["k2_stds"][r]))).to(x.device)
        got = M.data_fn(score_fn, xt, node[r, 0], node[r, 1], node[r, 2], dev)
        assert got.dtype == torch.float64
        assert np.array_equal(got.cpu().numpy(), cifar[f"k2_row{r}"])
    seq = [torch.from_numpy(a).to(dev) for a in cifar["k3_seq"]]
    got = M.weighted_sum(cifar["k3_coeff"], seq)
    assert got.dtype == torch.float32 and np.array_equal(got.cpu().numpy(), cifar["k3_out"])


def test_to_pixel_edges(dev):
    from naturaldiffusion_amd.CIFAR10NaturalInference import to_pixel, to_pixel_from_centered
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 3, 32, 32, generator=g) * 1.5
    x[0, 0, 0, :8] = torch.tensor([-1.0, 1.0, 0.0, -1.0000001, 1.0000001, 0.999999, -0.0, 0.00392157])
    assert np.array_equal(to_pixel_from_centered(x.to(dev)).numpy(), O.to_pixel(x).numpy())
    y = (x + 1.0) / 2.0
    want = torch.from_numpy(np.clip(y.permute(0, 2, 3, 1).numpy() * 255, 0, 255).astype(np.uint8))
    assert np.array_equal(to_pixel(y.to(dev)).numpy(), want.numpy())


def test_full_size_properties(dev, repo_root):
    """BASELINE config 2 size (B=512, 15 steps): dense == zero-skipped rows bit for bit, kernel ==
    oracle on a strided sample of elements, fast fp32 mode within 1e-5, linearity in the noise term."""
    from naturaldiffusion_amd.sampler import CifarNI
    C, B, node = O.load_coeff_npz(repo_root / "weights/step_15_weight_173.npz")
    Bn, E = 512, 512 * 3072
    g = torch.Generator().manual_seed(11)
    noise = torch.randn(E, generator=g)
    outs = [torch.randn(E, generator=g) for _ in range(15)]
    res = {}
    for tag, kw in (("sparse", {}), ("dense", dict(dense=True)), ("fast", dict(fast_f32=True))):
        ni = CifarNI(C, B, node, E, device=dev, **kw)
        x = noise.to(dev)
        nz = noise.to(dev)
        for k in range(15):
            x = ni.step(k, x, outs[k].to(dev), nz)
        res[tag] = x.cpu()
        if tag == "sparse":
            hist = ni.hist.cpu()
    assert torch.equal(res["sparse"], res["dense"])
    assert (res["fast"] - res["sparse"]).abs().max() <= 1e-5 * res["sparse"].abs().max()
    # oracle on every 997th element (pure elementwise recurrence -> any subset is exact)
    sel = torch.arange(0, E, 997)
    xs, hs = noise[sel], []
    for k in range(15):
        std = O.vp_std_f32(node[k, 0])
        hs.append(O.x0_from_score(xs, O.score_from_model_out(outs[k][sel], std), node[k, 1], node[k, 2]))
        xs = O.cifar_weighted_sum(C[k], hs) + noise[sel] * float(np.float32(B[k, 0]))
    assert torch.equal(res["sparse"][sel], xs)
    assert torch.equal(hist[14][sel], hs[14])


def test_argument_errors(dev):
    from naturaldiffusion_amd._lib import lib
    assert lib.natinf_step_f64hist(None, None, None, None, None, None, None, 0, 0.0, 0, 1.0, 1.0, 1.0, 0.0, 8, None) == -1
    t = torch.zeros(8, device=dev)
    h = torch.zeros(8, dtype=torch.float64, device=dev)
    p = lambda a: a.data_ptr()
    assert lib.natinf_step_f64hist(p(t), p(t), p(t), p(h), p(t), None, None, 0, 0.0, 0, 1.0, 1.0, 1.0, 0.0, 6, None) == -1
    assert lib.natinf_step_f64hist(p(t), p(t), p(t), p(h), p(t), None, None, 0, 0.0, 0, 1.0, 1.0, 1.0, 0.0, 8, None) == 0
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------ Validate form
class _FakeDiT:
    """the stand-in denoiser of tests/golden/make_golden.py (group validate)."""
    def __init__(self):
        self.base = O.analytic_eps_model()

    def forward(self, z, t, y):
        b = self.base(z, int(t[0]))
        is_null = bool((y == 1000).all())
        e = b * (0.9 if is_null else 1.1) + (0.0 if is_null else 0.02)
        return torch.cat([e, torch.zeros_like(e)], dim=1)


def test_validate_weighted_sum_mirror(dev, validate):
    from naturaldiffusion_amd import ValidateNaturalInference as V
    seq = [torch.from_numpy(a).to(dev) for a in validate["k3_seq"]]
    assert np.array_equal(V.weighted_sum(validate["k3_w"], seq).cpu().numpy(), validate["k3_out"])


@pytest.mark.parametrize("alg,key", [("ddpm_sympy", "ni_ddpm_sympy"), ("ddpm", "ni_ddpm"), ("ddim", "ni_ddim")])
def test_validate_natural_inference_bit_exact(dev, validate, monkeypatch, alg, key):
    from naturaldiffusion_amd import ValidateNaturalInference as V
    draws = [torch.from_numpy(validate["rng_z0"])] + [torch.from_numpy(a) for a in validate["rng_steps"]]
    it = iter(draws)
    monkeypatch.setattr(V.torch, "randn", lambda *a, **k: next(it).to(dev))
    monkeypatch.setattr(V.torch, "randn_like", lambda *a, **k: next(it).to(dev))
    monkeypatch.setattr(V, "denoiser_factory", lambda: _FakeDiT())
    monkeypatch.setattr(V, "device", "cuda:0")
    z = V.natural_inference(alg, 24)
    # the fixture is the VAE-decode input latents/0.18215; divide on the CPU (a GPU `tensor / python scalar`
    # multiplies by the reciprocal and can differ in the last ulp)
    assert np.array_equal((z.cpu() / 0.18215).numpy(), validate[key])


def test_validate_original_vs_natural(dev, validate, monkeypatch):
    """The reference's own check (Validate...:375-391), quantified: original sampler vs NI on the GPU path."""
    from naturaldiffusion_amd import ValidateNaturalInference as V
    monkeypatch.setattr(V, "denoiser_factory", lambda: _FakeDiT())
    monkeypatch.setattr(V, "device", "cuda:0")
    for orig, alg in ((V.ddpm_skip_sample, "ddpm_sympy"), (V.ddim_skip_sample, "ddim")):
        a = orig(24).clone()
        b = V.natural_inference(alg, 24)
        rel = ((a - b).abs().max() / a.abs().max()).item()
        assert rel < 5e-6, (alg, rel)          # fp32 tolerance proposed in SURVEY section 7 (observed 3e-7..6e-7)


# ------------------------------------------------------------------------------ SD3 form
def _fake_pipe():
    vel = O.analytic_velocity_model()

    class Sched:
        def set_timesteps(self, n, device=None):
            self.timesteps, self.sigmas = O.sd3_sigma_schedule(n)

    class Pipe:
        scheduler = Sched()

        def encode_prompt(self, prompt, **k):
            return ("T", "N", "PT", "PN")

        def transformer(self, hidden_states, timestep, encoder_hidden_states, pooled_projections, return_dict=False):
            return [vel(hidden_states, timestep[0], encoder_hidden_states == "T")]
    return Pipe()


def _scaled(z):
    return (z / 1.5305) + 0.0609


def test_sd3_weighted_sum_mirror(dev, sd3):
    from naturaldiffusion_amd import SD3NaturalInference as S
    seq = [torch.from_numpy(a).to(dev) for a in sd3["k3_seq"]]
    assert np.array_equal(S.weighted_sum(seq, sd3["k3_W"]).cpu().numpy(), sd3["k3_out"])
    assert np.array_equal(S.weighted_sum(seq, None).cpu().numpy(), sd3["k3_out_uniform"])


def test_sd3_natural_inference_bit_exact(dev, sd3):
    from naturaldiffusion_amd import SD3NaturalInference as S
    noises = torch.from_numpy(sd3["noises"]).to(dev)
    finals = S.sd_natural_inference_tx(pipe=_fake_pipe(), device="cuda:0", noises=noises, decode=False)
    assert np.array_equal(_scaled(finals[0].cpu()).numpy(), sd3["final_plain_scaled"])
    assert np.array_equal(_scaled(finals[1].cpu()).numpy(), sd3["final_sharp_scaled"])
    out = S.sd_euler_natural_inference_tx(pipe=_fake_pipe(), device="cuda:0", noises=noises, decode=False)
    assert np.array_equal(_scaled(out.cpu()).numpy(), sd3["final_euler_ni_scaled"])


def test_sd3_full_size_step_matches_oracle(dev, repo_root):
    """one step at the real latent size (4x16x128x128) with a long history: kernel == oracle, bit for bit."""
    from naturaldiffusion_amd.sampler import SD3NI
    W = O.load_sd3_csv(repo_root / "weights/sd3_step_28_weight.csv")
    _, sigmas = O.sd3_sigma_schedule(28)
    E = 4 * 16 * 128 * 128
    g = torch.Generator().manual_seed(2)
    k = 20
    hist = [(torch.randn(E, generator=g) * 1.3).half() for _ in range(k)]
    x, vt, vn, nz = [(torch.randn(E, generator=g)).half() for _ in range(4)]
    ni = SD3NI(W, sigmas, E, device=dev)
    for j in range(k):
        ni.hist[j].copy_(hist[j])
    mean, xn = ni.step(k, x.to(dev), vt.to(dev), vn.to(dev), nz.to(dev))
    sig = sigmas[k]
    x0n = x - sig * vn
    x0t = x - sig * vt
    f = x0n + 7.0 * (x0t - x0n)
    want_mean = O.sd3_weighted_mean(hist + [f], W)
    want_next = sigmas[k + 1] * nz + (1 - sigmas[k + 1]) * want_mean
    assert torch.equal(ni.hist[k].cpu(), f)
    assert torch.equal(mean.cpu(), want_mean)
    assert torch.equal(xn.cpu(), want_next)


# ------------------------------------------------------------------------------ edge cases
def test_long_stochastic_history_matches_oracle(dev):
    """a 120-step matrix with a dense lower triangle AND a dense noise matrix (every column used) in the
    Validate form: the term loops run far past their unroll factor; kernel == oracle on every element."""
    from naturaldiffusion_amd.sampler import ValidateNI
    rs = np.random.RandomState(7)
    N, E = 120, 4 * 1024
    C = np.tril(rs.randn(N, N) * 0.2)
    Bm = np.zeros((N, N + 1))
    for k in range(N):
        Bm[k, :k + 2] = rs.randn(k + 2) * 0.1
    C[5, 2] = 0.0; Bm[9, 3] = 0.0; C[17, 17] = 0.0                     # zero entries incl. a zero diagonal
    node = np.zeros((N + 1, 3))
    c1 = rs.rand(N) + 0.5; c2 = rs.rand(N) * 0.5
    g = torch.Generator().manual_seed(3)
    z0 = torch.randn(E, generator=g)
    eps = [torch.randn(E, generator=g) for _ in range(N)]
    noi = [torch.randn(E, generator=g) for _ in range(N + 1)]
    for dense in (False, True):
        ni = ValidateNI(C, Bm, node, c1.astype(np.float32), c2.astype(np.float32), E, device=dev, dense=dense)
        for j in range(N + 1):
            ni.hist_eps[j].copy_(noi[j])
        z = z0.to(dev)
        seq_x0, zo = [], z0
        for k in range(N):
            z = ni.step(k, z, eps[k].to(dev), None, 0.0)
            x0 = float(np.float32(c1[k])) * zo - float(np.float32(c2[k])) * eps[k]
            seq_x0.append(x0)
            zo = O.validate_weighted_sum(C[k], seq_x0) + O.validate_weighted_sum(Bm[k, :k + 2], noi[:k + 2])
            if k % 40 == 39 or k == N - 1:
                assert torch.equal(z.cpu(), zo), f"step {k} (dense={dense})"
        assert torch.isfinite(zo).all()


def test_degenerate_rows_and_sizes(dev):
    """single-step matrix (no history terms at all), the smallest legal E, and E not a multiple of the vector width."""
    from naturaldiffusion_amd.sampler import CifarNI
    C = np.array([[0.75]]); Bm = np.array([[0.5]]); node = np.array([[1.0, 0.3, 0.9], [0.0, 1.0, 0.0]])
    for E in (4, 8, 4 * 1031):
        g = torch.Generator().manual_seed(E)
        x, out = torch.randn(E, generator=g), torch.randn(E, generator=g)
        ni = CifarNI(C, Bm, node, E, device=dev, stds=[0.97])
        got = ni.step(0, x.to(dev), out.to(dev), x.to(dev)).cpu()
        x0 = O.x0_from_score(x, O.score_from_model_out(out, torch.tensor(0.97)), 0.3, 0.9)
        want = O.cifar_weighted_sum(C[0], [x0]) + x * 0.5
        assert torch.equal(got, want)
    with pytest.raises(ValueError):
        CifarNI(C, Bm, node, 6, device=dev)
    # non-finite data: dense rows reproduce the reference's NaN propagation (inf * 0 = nan), sparse rows do not
    C2 = np.array([[1.0, 0.0], [0.0, 1.0]]); B2 = np.zeros((2, 2)); node2 = np.array([[1.0, 1.0, 0.0], [0.5, 1.0, 0.0], [0.0, 1.0, 0.0]])
    x = torch.zeros(4); bad = torch.tensor([float("inf"), 1.0, 2.0, 3.0])
    for dense, expect_nan in ((True, True), (False, False)):
        ni = CifarNI(C2, B2, node2, 4, device=dev, dense=dense, stds=[1.0, 1.0])
        ni.step(0, bad.to(dev), x.to(dev), x.to(dev))          # x0_0 = [inf, 1, 2, 3]
        r = ni.step(1, x.to(dev), x.to(dev), x.to(dev)).cpu()  # row 1 = [0, 1]: inf*0 term present only when dense
        assert bool(torch.isnan(r[0])) == expect_nan
