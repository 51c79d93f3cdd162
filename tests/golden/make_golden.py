#!/usr/bin/env python3
"""Capture golden vectors from the REFERENCE itself (build container only).

Runs ``/root/reference``'s own Python on CPU under ``sys.modules`` stubs for the
packages this image lacks (recipes: SURVEY.md section 8c) and writes small ``.npz``
fixtures next to this file.  The reference never travels: the GPU box only sees the
fixtures.  Usage::

    python tests/golden/make_golden.py            # all groups, one subprocess each
    python tests/golden/make_golden.py cifar      # one group

Groups
  loaders   K1  shape / dtype / nnz / checksum table of every shipped coefficient file
  cifar     K2-K4  data_fn, weighted_sum, full NI trajectories (CIFAR10NaturalInference.py)
  validate  K3/K5  weighted_sum, ddpm/ddim originals vs natural_inference (ValidateNaturalInference.py)
  sd3       K3/K6  weighted_sum, sd_natural_inference_tx, sd_euler (SD3NaturalInference.py)
  ncsnpp    K7  NCSNpp forward (reference nn.Module) on the oracle's synthetic weights
  k5        K5  the vendored classical solver itself (deps/dpm_solver_pytorch.py:906 singlestep updates) next to NI with
                the shipped dpmsolverpp2s_018 / dpmsolver2s_018 / dpmsolver3s_018 matrices, same noise, same denoiser
  ckpt      B-iii  a score_sde checkpoint written through the reference's own ExponentialMovingAverage.state_dict()
                and DataParallel state_dict() (utils.py:22-29, ema.py:91-97): the file and what restore + ema.copy_to give
"""
import hashlib
import json
import os
import subprocess
import sys
import types
from pathlib import Path

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(REPO))


# --------------------------------------------------------------------------- #
def _stub(name, **attrs):
    import importlib.machinery
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)      # torch._dynamo walks sys.modules with find_spec
    sys.modules[name] = m
    return m


class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _stub_cifar_env():
    _stub("ml_collections", ConfigDict=_AttrDict)
    jax = _stub("jax")
    jax.random = _stub("jax.random")
    jax.numpy = _stub("jax.numpy")
    _stub("cv2")
    _stub("tensorflow")
    _stub("tensorflow_datasets")
    _stub("scienceplots")
    pf = _stub("pytorch_fid")
    pf.inception = _stub("pytorch_fid.inception", InceptionV3=object)
    pf.fid_score = _stub("pytorch_fid.fid_score", calculate_frechet_distance=None)
    _stub("th_deis")
    _stub("op", FusedLeakyReLU=None, fused_leaky_relu=None, upfirdn2d=None)
    sys.path.insert(0, str(REF / "src"))


def _sha(a):
    import numpy as np
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


# --------------------------------------------------------------------------- #
def group_loaders():
    import numpy as np
    import pandas as pd
    table = {}
    files = sorted((REF / "weights").glob("*.npz")) + sorted((REF / "results").glob("*/*.npz"))
    for f in files:
        C, B, node = np.load(f).values()           # the reference's positional read
        table[str(f.relative_to(REF))] = dict(
            C=list(C.shape), B=list(B.shape), node=list(node.shape),
            dtype=[str(C.dtype), str(B.dtype), str(node.dtype)],
            nnzC=int(np.count_nonzero(C)), nnzB=int(np.count_nonzero(B)),
            shaC=_sha(C), shaB=_sha(B), shaN=_sha(node))
    for f in sorted((REF / "weights").glob("*.csv")):
        W = pd.read_csv(f, index_col=0).to_numpy()
        table[str(f.relative_to(REF))] = dict(W=list(W.shape), dtype=[str(W.dtype)],
                                              nnz=int(np.count_nonzero(W)), sha=_sha(W),
                                              rowsum_last=float(W[-1].sum()))
    (HERE / "k1_loaders.json").write_text(json.dumps(table, indent=1, sort_keys=True))
    print("loaders:", len(table), "files")


def group_cifar():
    import numpy as np
    import torch
    _stub_cifar_env()
    import CIFAR10NaturalInference as R
    from oracle import ni_oracle as O

    model_fn = O.analytic_vp_model()
    sde = R.VPSDE(beta_min=0.1, beta_max=20.0, N=1000)
    score_fn = R.mutils.get_score_fn(sde, model_fn_module(model_fn), train=False, continuous=True)

    out = {}
    g = torch.Generator().manual_seed(1234)

    def stds_of(node):          # the fp32 std the reference's score_fn used on THIS host (sde_lib.py:141-145)
        return np.array([float(sde.marginal_prob(torch.zeros(1), torch.ones(1) * t)[1][0]) for t in node[:-1, 0]], np.float32)
    # K2 data_fn at three rows of step_15
    C, B, node = np.load(REF / "weights/step_15_weight_173.npz").values()
    xt = torch.randn(2, 3, 32, 32, generator=g)
    out["k2_xt"] = xt.numpy()
    out["k2_stds"] = stds_of(node)
    for r in (0, 7, 14):
        out[f"k2_row{r}"] = R.data_fn(score_fn, xt, node[r, 0], node[r, 1], node[r, 2], "cpu").numpy()
    # K3 weighted_sum incl. negative / zero coefficients
    seq = [torch.randn(2, 3, 8, 8, generator=g, dtype=torch.float64) for _ in range(6)]
    coeff = np.array([0.3, -1.25, 0.0, 2.5e-3, 0.0, 1.75])
    out["k3_seq"] = np.stack([s.numpy() for s in seq])
    out["k3_coeff"] = coeff
    out["k3_out"] = R.weighted_sum(coeff, seq).numpy()
    # K4 full trajectories; the loop body is the reference's lines 292-304 driven through
    # the reference's own data_fn / weighted_sum (natural_inference_tx itself asserts on
    # the missing checkpoint and generates 50k samples)
    for name in ("step_5_weight_00", "step_10_weight_42", "step_15_weight_173"):
        C, B, node = np.load(REF / f"weights/{name}.npz").values()
        noise = torch.randn(2, 3, 32, 32, generator=g)
        ts = node[:, 0]
        seq_x0, xs, x = [], [noise], noise
        for kk in range(ts.shape[0] - 1):
            seq_x0.append(R.data_fn(score_fn, x, ts[kk], node[kk, 1], node[kk, 2], "cpu"))
            x = R.weighted_sum(C[kk], seq_x0) + B[kk, 0] * noise
            xs.append(x)
        out[f"k4_{name}_xs"] = np.stack([t.numpy() for t in xs])
        out[f"k4_{name}_stds"] = stds_of(node)
        out[f"k4_{name}_pix"] = R.to_pixel(R.datasets.get_data_inverse_scaler(_AttrDict(data=_AttrDict(centered=True)))(x)).numpy()
    # K5 (continuous grid): coefficient-matrix equivalents of classical samplers shipped under results/
    for rel in ("dpmsolverpp/dpmsolverpp2s_018", "euler_heun/ode_euler_018"):
        C, B, node = np.load(REF / f"results/{rel}.npz").values()
        noise = torch.randn(2, 3, 32, 32, generator=g)
        ts = node[:, 0]
        seq_x0, x = [], noise
        for kk in range(ts.shape[0] - 1):
            seq_x0.append(R.data_fn(score_fn, x, ts[kk], node[kk, 1], node[kk, 2], "cpu"))
            x = R.weighted_sum(C[kk], seq_x0) + B[kk, 0] * noise
        key = rel.split("/")[1]
        out[f"k5_{key}_noise"] = noise.numpy()
        out[f"k5_{key}_stds"] = stds_of(node)
        out[f"k5_{key}_final"] = x.numpy()
    np.savez_compressed(HERE / "cifar_form.npz", **out)
    print("cifar:", len(out), "arrays")


def model_fn_module(fn):
    """wrap a callable as the nn.Module-like object get_model_fn expects."""
    class M:
        def eval(self):
            return self

        def train(self):
            return self

        def __call__(self, x, labels):
            return fn(x, labels)
    return M()


def group_validate():
    import numpy as np
    import torch
    import shutil, tempfile
    dm = _stub("diffusers")
    dm.models = _stub("diffusers.models", AutoencoderKL=None)
    tm = _stub("timm"); tm.models = _stub("timm.models")
    tm.models.vision_transformer = _stub("timm.models.vision_transformer", PatchEmbed=object, Attention=object, Mlp=object)
    tv = _stub("torchvision"); tv.utils = _stub("torchvision.utils", save_image=None)
    sys.path.insert(0, str(REF / "src"))
    import ValidateNaturalInference as V
    from oracle import ni_oracle as O

    # ---- redirect the hard-coded "cuda:0" to CPU
    def _cpuify(fn):
        def w(*a, **k):
            if "device" in k and k["device"] is not None:
                k["device"] = "cpu"
            return fn(*a, **k)
        return w
    for nm in ("randn", "ones", "tensor", "randn_like", "zeros"):
        setattr(torch, nm, _cpuify(getattr(torch, nm)))
    _orig_to = torch.Tensor.to
    def _to(self, *a, **k):
        a = tuple("cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a)
        if isinstance(k.get("device"), str):
            k["device"] = "cpu"
        return _orig_to(self, *a, **k)
    torch.Tensor.to = _to

    eps_fn = O.analytic_eps_model()

    class FakeDiT:
        """forward(z, t, y) -> 8 channels; cond / uncond differ so CFG is exercised."""
        def to(self, *_a, **_k): return self
        def eval(self): return self
        def load_state_dict(self, *_a, **_k): return None
        def forward(self, z, t, y):
            base = eps_fn(z, int(t[0]))
            is_null = bool((y == 1000).all())
            e = base * (0.9 if is_null else 1.1) + (0.0 if is_null else 0.02)
            return torch.cat([e, torch.zeros_like(e)], dim=1)

    captured = {}
    class FakeVAE:
        @staticmethod
        def from_pretrained(_p): return FakeVAE()
        def to(self, *_a, **_k): return self
        def eval(self): return self
        def decode(self, z):
            captured["z"] = z.clone()          # = final latents / 0.18215 (Validate...:254)
            return types.SimpleNamespace(sample=z)
    V.DiT_models = {"DiT-XL/2": lambda **k: FakeDiT()}
    V.AutoencoderKL = FakeVAE
    V.save_image = lambda *a, **k: None
    torch.load = lambda *a, **k: {}
    tmp = Path(tempfile.mkdtemp())
    shutil.copytree(REF / "results", tmp / "results")
    # ddim_sympy_024 is not shipped; it equals ddim_024 in C,B
    C, B, node = np.load(REF / "results/ddim/ddim_024.npz").values()
    node = node.copy(); node[0, 1] = 0.0064
    np.savez(tmp / "results/ddim/ddim_sympy_024.npz", past_xstart_coeff=C, past_epsilon_coeff=B, node_coeff=node)
    V.root_path = tmp
    V.vae_path, V.model_path = "x", "y"

    out = {}
    g = torch.Generator().manual_seed(77)
    seq = [torch.randn(2, 4, 8, 8, generator=g) for _ in range(5)]
    w = np.array([0.5, -0.125, 0.0, 3.0e-2, 1.0 / 3.0])
    out["k3_seq"] = np.stack([s.numpy() for s in seq]); out["k3_w"] = w
    out["k3_out"] = V.weighted_sum(w, seq).numpy()

    def run(fn, *a):
        fn(*a)
        return captured["z"].numpy()
    out["ddpm_original"] = run(V.ddpm_skip_sample, 24)
    out["ddim_original"] = run(V.ddim_skip_sample, 24)
    out["ni_ddpm_sympy"] = run(V.natural_inference, "ddpm_sympy", 24)
    out["ni_ddpm"] = run(V.natural_inference, "ddpm", 24)
    out["ni_ddim"] = run(V.natural_inference, "ddim", 24)
    out["ni_ddim_sympy"] = run(V.natural_inference, "ddim_sympy", 24)
    # the RNG draws the scripts consumed (seed 0: initial randn, then one randn_like per step)
    torch.manual_seed(0)
    out["rng_z0"] = torch.randn(8, 4, 32, 32).numpy()
    out["rng_steps"] = np.stack([torch.randn(8, 4, 32, 32).numpy() for _ in range(24)])
    np.savez_compressed(HERE / "validate_form.npz", **out)
    shutil.rmtree(tmp)
    for k in ("ni_ddpm_sympy", "ni_ddpm"):
        d = np.abs(out[k] - out["ddpm_original"]).max() / np.abs(out["ddpm_original"]).max()
        print(f"validate: {k} vs original rel {d:.3e}")
    d = np.abs(out["ni_ddim"] - out["ddim_original"]).max() / np.abs(out["ddim_original"]).max()
    print(f"validate: ni_ddim vs original rel {d:.3e}")


def group_sd3():
    import numpy as np
    import torch
    from oracle import ni_oracle as O
    LAT = 16                                   # shrink 128x128 latents to 16x16
    vel = O.analytic_velocity_model()
    timesteps, sigmas = O.sd3_sigma_schedule(28)
    captured = []

    class Sched:
        def set_timesteps(self, n, device=None):
            self.timesteps, self.sigmas = O.sd3_sigma_schedule(n)
    class Pipe:
        scheduler = Sched()
        vae = types.SimpleNamespace(config=types.SimpleNamespace(scaling_factor=1.5305, shift_factor=0.0609),
                                    decode=lambda z, return_dict=False: (captured.append(z.clone()) or [z])[0:1])
        image_processor = types.SimpleNamespace(postprocess=lambda imgs, output_type="pil": [np.zeros((2, 2, 3), np.uint8)] * len(imgs))
        @staticmethod
        def from_pretrained(*a, **k): return Pipe()
        def to(self, *_a, **_k): return self
        def encode_prompt(self, prompt, **k): return ("T", "N", "PT", "PN")
        def transformer(self, hidden_states, timestep, encoder_hidden_states, pooled_projections, return_dict=False):
            return [vel(hidden_states, timestep[0], encoder_hidden_states == "T")]
    _stub("diffusers", StableDiffusion3Pipeline=Pipe)
    _stub("cv2", imwrite=lambda *a, **k: True, resize=lambda img, size, interpolation=None: np.zeros((size[1], size[0], 3), np.uint8), INTER_NEAREST=0)
    sys.path.insert(0, str(REF / "src"))
    _randn, _Gen = torch.randn, torch.Generator
    def randn(*shape, **k):
        k.pop("generator", None); k["device"] = "cpu"
        shape = tuple(LAT if s == 128 else s for s in shape)
        return _randn(*shape, generator=_Gen().manual_seed(10), **k)
    torch.randn = randn
    torch.Generator = lambda *_a, **_k: types.SimpleNamespace(manual_seed=lambda s: None)
    _to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: _to(self, *tuple("cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a), **k)
    import SD3NaturalInference as S
    import tempfile
    S.root_path = Path(tempfile.mkdtemp()); os.makedirs(S.root_path / "results/sd3");
    os.symlink(REF / "weights", S.root_path / "weights")

    out = {}
    g = torch.manual_seed(5)
    seq = [(_randn(2, 4, 8, 8) * 2).half() for _ in range(6)]
    W = np.tril(np.abs(np.random.RandomState(0).randn(6, 6)).round(2)); W[5, 2] = 0.0
    out["k3_seq"] = np.stack([s.numpy() for s in seq]); out["k3_W"] = W
    out["k3_out"] = S.weighted_sum(seq, W).numpy()
    out["k3_out_uniform"] = S.weighted_sum(seq, None).numpy()

    out["noises"] = randn(4, 16, 128, 128, dtype=torch.float16).numpy()
    S.sd_natural_inference_tx()
    # per CSV: 28*5*4/12 chunk decodes then one final decode; finals are the un-chunked ones
    finals = [c for c in captured if c.shape[0] == 4]
    assert len(finals) == 2, [c.shape for c in captured]
    unscale = lambda z: ((z - 0.0609) * 1.5305)
    out["final_plain_scaled"] = finals[0].numpy(); out["final_sharp_scaled"] = finals[1].numpy()
    captured.clear()
    S.sd_euler_natural_inference_tx()
    out["final_euler_ni_scaled"] = [c for c in captured if c.shape[0] == 4][-1].numpy()
    out["sigmas"] = sigmas.numpy(); out["timesteps"] = timesteps.numpy()
    np.savez_compressed(HERE / "sd3_form.npz", **out)
    print("sd3:", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


def group_ncsnpp():
    import numpy as np
    import torch
    _stub_cifar_env()
    import CIFAR10NaturalInference as R
    from oracle import ncsnpp_oracle as N
    torch.set_num_threads(8)
    config = R.configs.get_config()
    config.device = torch.device("cpu")
    model = R.mutils.create_model(config)              # DataParallel(NCSNpp)
    net = model.module
    P = N.make_params(seed=0)
    sd = dict(P); sd["sigmas"] = net.sigmas
    missing = net.load_state_dict(sd, strict=True)
    n_param = sum(p.numel() for p in net.parameters())
    net.eval()
    taps = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, k=k: taps.__setitem__(k, o.detach())) for k, m in enumerate(net.all_modules)]
    g = torch.Generator().manual_seed(4321)
    x = torch.randn(2, 3, 32, 32, generator=g)
    labels = torch.tensor([0.65016 * 999, 0.05076 * 999], dtype=torch.float32)
    with torch.no_grad():
        y = net(x, labels)
    for h in hooks:
        h.remove()
    out = dict(x=x.numpy(), labels=labels.numpy(), y=y.numpy(), n_param=np.int64(n_param))
    for k, t in taps.items():
        t = t.float()
        out[f"tap{k:02d}_stats"] = np.array([t.mean().item(), t.std().item(), t.abs().max().item()], np.float64)
        out[f"tap{k:02d}_head"] = t.flatten()[:32].numpy()
        out[f"tap{k:02d}_shape"] = np.array(t.shape, np.int64)
    # full activations: down res-block, channel-changing, attention @16 and @4, concat-input, up, 384-ch concat
    for k in (7, 8, 9, 27, 29, 34, 45):
        out[f"tap{k:02d}_full"] = taps[k].numpy()
    np.savez_compressed(HERE / "ncsnpp_forward.npz", **out)
    print("ncsnpp: params", n_param, "| y absmax", float(np.abs(out["y"]).max()), "| taps", len(taps))


def group_dit():
    """deps/DiT/models.py's own DiT class (timm's PatchEmbed / Attention / Mlp stubbed by their published
    definitions) on the oracle's synthetic weights: two small configurations (head_dim 64 and 72) in full, DiT-XL/2
    as output statistics."""
    import numpy as np
    import torch
    import torch.nn as nn

    class PatchEmbed(nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, bias=True):
            super().__init__()
            self.patch_size = (patch_size, patch_size)
            self.num_patches = (img_size // patch_size) ** 2
            self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)

        def forward(self, x):
            return self.proj(x).flatten(2).transpose(1, 2)

    class Attention(nn.Module):
        def __init__(self, dim, num_heads=8, qkv_bias=False):
            super().__init__()
            self.num_heads, self.scale = num_heads, (dim // num_heads) ** -0.5
            self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
            self.proj = nn.Linear(dim, dim)

        def forward(self, x):
            B, N, C = x.shape
            qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
            q, k, v = qkv.unbind(0)
            attn = ((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1)
            return self.proj((attn @ v).transpose(1, 2).reshape(B, N, C))

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, act_layer=nn.GELU, drop=0.0):
            super().__init__()
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features, in_features)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    tm = _stub("timm"); tm.models = _stub("timm.models")
    tm.models.vision_transformer = _stub("timm.models.vision_transformer", PatchEmbed=PatchEmbed, Attention=Attention, Mlp=Mlp)
    sys.path.insert(0, str(REF / "deps/DiT"))
    import models as M
    from oracle import dit_oracle as D
    torch.set_num_threads(8)
    out = {}
    g = torch.Generator().manual_seed(99)
    for tag, depth, hid, heads, B in (("s64", 2, 128, 2, 3), ("s72", 1, 576, 8, 2), ("xl2", 28, 1152, 16, 1)):
        net = M.DiT(depth=depth, hidden_size=hid, num_heads=heads).eval()
        P = D.make_params(depth, hid, seed=7)
        net.load_state_dict(P, strict=True)
        x = torch.randn(B, 4, 32, 32, generator=g)
        t = torch.tensor([999.0, 500.0, 41.0][:B])
        y = torch.tensor([207, 1000, 3][:B])
        with torch.no_grad():
            r = net(x, t, y)
        out[f"{tag}_x"], out[f"{tag}_t"], out[f"{tag}_y"] = x.numpy(), t.numpy(), y.numpy()
        if tag == "xl2":
            out[f"{tag}_stats"] = np.array([r.mean().item(), r.std().item(), r.abs().max().item()])
            out[f"{tag}_head"] = r.flatten()[:256].numpy()
            out[f"{tag}_nparam"] = np.int64(sum(p.numel() for p in net.parameters()))
        else:
            out[f"{tag}_out"] = r.numpy()
    np.savez_compressed(HERE / "dit_forward.npz", **out)
    print("dit:", {k: getattr(v, "shape", v) for k, v in out.items() if k.endswith(("out", "stats", "nparam"))})


def group_k5():
    """DPM_Solver.singlestep_dpm_solver_update over linspace(1, 1e-3, K+1) == CIFAR10-form NI with the shipped
    matrix of the same sampler.  ``orig`` = the reference's solver class, ``ni`` = the reference's data_fn / weighted_sum
    loop (src/CIFAR10NaturalInference.py:292-304); both on the analytic denoiser, fp32 model I/O."""
    import numpy as np
    import torch
    _stub_cifar_env()
    import CIFAR10NaturalInference as R
    from oracle import ni_oracle as O
    model_fn = O.analytic_vp_model()
    sde = R.VPSDE(beta_min=0.1, beta_max=20.0, N=1000)
    score_fn = R.mutils.get_score_fn(sde, model_fn_module(model_fn), train=False, continuous=True)
    noise_fn = lambda x, t: model_fn(x, t * 999)               # the reference's get_noise_fn contract: eps-hat at continuous t
    ns = R.NoiseScheduleVP('linear', continuous_beta_0=sde.beta_0, continuous_beta_1=sde.beta_1)
    out = {}
    g = torch.Generator().manual_seed(4242)
    cases = (("dpmsolverpp/dpmsolverpp2s_018", "dpmsolver++", 2, 9), ("dpmsolver/dpmsolver2s_018", "dpmsolver", 2, 9),
             ("dpmsolver/dpmsolver3s_018", "dpmsolver", 3, 6))
    for rel, alg, order, K in cases:
        key = rel.split("/")[1]
        C, B, node = np.load(REF / f"results/{rel}.npz").values()
        noise = torch.randn(2, 3, 32, 32, generator=g)
        solver = R.DPM_Solver(noise_fn, ns, algorithm_type=alg)
        grid = torch.linspace(1.0, 1e-3, K + 1)
        x = noise
        with torch.no_grad():
            for i in range(K):
                kw = dict(r1=0.5) if order == 2 else dict(r1=1.0 / 3.0, r2=2.0 / 3.0)
                x = solver.singlestep_dpm_solver_update(x, grid[i:i + 1], grid[i + 1:i + 2], order, solver_type='dpmsolver', **kw)
        ts = node[:, 0]
        seq_x0, y = [], noise
        for kk in range(ts.shape[0] - 1):
            seq_x0.append(R.data_fn(score_fn, y, ts[kk], node[kk, 1], node[kk, 2], "cpu"))
            y = R.weighted_sum(C[kk], seq_x0) + B[kk, 0] * noise
        out[f"{key}_noise"] = noise.numpy()
        out[f"{key}_stds"] = np.array([float(sde.marginal_prob(torch.zeros(1), torch.ones(1) * t)[1][0]) for t in node[:-1, 0]], np.float32)
        out[f"{key}_orig"] = x.numpy()
        out[f"{key}_ni"] = y.numpy()
        print(f"k5: {key}: |orig - ni| max {np.abs(x.numpy() - y.numpy()).max():.3e}  (|orig| max {np.abs(x.numpy()).max():.3f})")
    np.savez_compressed(HERE / "k5_classical.npz", **out)


def group_ddim_vp():
    """BASELINE config 3 names "DDIM / DPMSolver++ coeff-matrix equivalents": DDIM on the CONTINUOUS VP grid has no shipped matrix
    (results/ddim/* are the discrete DiT schedule), so it comes from naturaldiffusion_amd.coeffgen.ddim_vp_continuous.  Pinned here
    against the vendored solver itself: ``orig`` / ``orig_pp`` = DPM_Solver.dpm_solver_first_update (deps/dpm_solver_pytorch.py:547-592,
    "DPM-Solver-1 (equivalent to DDIM)") in its noise-prediction and data-prediction forms over the time grid; ``ni`` = the
    reference's data_fn / weighted_sum loop (src/CIFAR10NaturalInference.py:292-304) with the generated matrix, same noise, same
    analytic denoiser.  Two grids: linspace(1, 1e-3, 19) (the solver's uniform-time grid) and the quadratic 15-step grid of the
    shipped weights/step_15_*.npz."""
    import numpy as np
    import torch
    _stub_cifar_env()
    import CIFAR10NaturalInference as R
    from oracle import ni_oracle as O
    from naturaldiffusion_amd import coeffgen as G
    model_fn = O.analytic_vp_model()
    sde = R.VPSDE(beta_min=0.1, beta_max=20.0, N=1000)
    score_fn = R.mutils.get_score_fn(sde, model_fn_module(model_fn), train=False, continuous=True)
    noise_fn = lambda x, t: model_fn(x, t * 999)
    ns = R.NoiseScheduleVP('linear', continuous_beta_0=sde.beta_0, continuous_beta_1=sde.beta_1)
    out = {}
    g = torch.Generator().manual_seed(777)
    for key, ts in (("lin18", np.linspace(1.0, 1e-3, 19)), ("quad15", G.quadratic_time_grid(15))):
        C, B, node = G.ddim_vp_continuous(ts)
        noise = torch.randn(2, 3, 32, 32, generator=g)
        grid = torch.from_numpy(np.asarray(ts, np.float64)).float()
        res = {}
        for tag, alg in (("orig", "dpmsolver"), ("orig_pp", "dpmsolver++")):
            solver = R.DPM_Solver(noise_fn, ns, algorithm_type=alg)
            x = noise
            with torch.no_grad():
                for i in range(len(ts) - 1):
                    x = solver.dpm_solver_first_update(x, grid[i:i + 1], grid[i + 1:i + 2])
            res[tag] = x.numpy()
        seq_x0, y = [], noise
        for kk in range(node.shape[0] - 1):
            seq_x0.append(R.data_fn(score_fn, y, node[kk, 0], node[kk, 1], node[kk, 2], "cpu"))
            y = R.weighted_sum(C[kk], seq_x0) + B[kk, 0] * noise
        out[f"{key}_ts"], out[f"{key}_C"], out[f"{key}_B"], out[f"{key}_node"] = np.asarray(ts, np.float64), C, B, node
        out[f"{key}_noise"] = noise.numpy()
        out[f"{key}_stds"] = np.array([float(sde.marginal_prob(torch.zeros(1), torch.ones(1) * t)[1][0]) for t in node[:-1, 0]], np.float32)
        out[f"{key}_orig"], out[f"{key}_orig_pp"], out[f"{key}_ni"] = res["orig"], res["orig_pp"], y.numpy()
        print(f"ddim_vp: {key}: |orig - ni| max {np.abs(res['orig'] - y.numpy()).max():.3e}, |orig_pp - ni| max {np.abs(res['orig_pp'] - y.numpy()).max():.3e}"
              f"  (|orig| max {np.abs(res['orig']).max():.3f})")
    np.savez_compressed(HERE / "ddim_vp.npz", **out)


def group_ddpm():
    """The reference's ``DDPM`` class (deps/score_sde_pytorch/models/ddpm.py:39-181, registered as 'ddpm'; configuration
    configs/vp/ddpm/cifar10_continuous.py) on the synthetic weights of oracle/ddpm_oracle.py: the network of the checkpoint the
    reference's docstring names (src/CIFAR10NaturalInference.py:416).  Output, statistics / head of every module output, a few
    activations in full (a width-changing res-block with its NIN shortcut, Downsample, attention, a concat-input block, Upsample)."""
    import numpy as np
    import torch
    _stub_cifar_env()
    import CIFAR10NaturalInference as R                      # brings models.utils / layers onto the path
    from configs.vp.ddpm import cifar10_continuous as ddpm_cfg
    from models import ddpm as ddpm_mod                      # registers 'ddpm'
    from oracle import ddpm_oracle as D
    torch.set_num_threads(8)
    config = ddpm_cfg.get_config()
    config.device = torch.device("cpu")
    net = R.mutils.create_model(config).module
    assert type(net).__name__ == "DDPM"
    P = D.make_params(seed=0)
    sd = dict(P); sd["sigmas"] = net.sigmas
    net.load_state_dict(sd, strict=True)
    n_param = sum(p.numel() for p in net.parameters())
    net.eval()
    taps = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, k=k: taps.__setitem__(k, o.detach())) for k, m in enumerate(net.all_modules)]
    g = torch.Generator().manual_seed(8765)
    x = torch.randn(2, 3, 32, 32, generator=g)
    labels = torch.tensor([0.81 * 999, 0.0123 * 999], dtype=torch.float32)
    with torch.no_grad():
        y = net(x, labels)
    for h in hooks:
        h.remove()
    out = dict(x=x.numpy(), labels=labels.numpy(), y=y.numpy(), n_param=np.int64(n_param), n_modules=np.int64(len(net.all_modules)),
               names=np.array([n for n, _ in net.named_parameters()]))
    for k, t in taps.items():
        t = t.float()
        out[f"tap{k:02d}_stats"] = np.array([t.mean().item(), t.std().item(), t.abs().max().item()], np.float64)
        out[f"tap{k:02d}_head"] = t.flatten()[:32].numpy()
        out[f"tap{k:02d}_shape"] = np.array(t.shape, np.int64)
    kinds = {m.idx: m.kind for m in D.plan()}
    full = [next(k for k in kinds if kinds[k] == "down"), next(k for k in kinds if kinds[k] == "attn"), next(k for k in kinds if kinds[k] == "up")]
    full += [m.idx for m in D.plan() if m.kind == "res" and m.cin != m.cout][:2] + [m.idx for m in D.plan() if m.kind == "res" and m.cin > 256][:1]
    for k in sorted(set(full)):
        out[f"tap{k:02d}_full"] = taps[k].numpy()
    np.savez_compressed(HERE / "ddpm_forward.npz", **out)
    print("ddpm: params", n_param, "| modules", len(net.all_modules), "| y absmax", float(np.abs(out["y"]).max()), "| full taps", sorted(set(full)))


def group_ckpt():
    """A score_sde checkpoint made the way the reference's training loop makes one (utils.save_checkpoint, utils.py:22-29):
    {'optimizer', 'model' (DataParallel state_dict, 'module.' keys), 'ema' (ExponentialMovingAverage.state_dict(),
    ema.py:91-97), 'step'}, at width nf = 8 so that the reference-written pickle is a 2 MB fixture (the full model is
    61.8 M floats x 2).  The EMA shadow parameters differ from the model's -- they are what sampling uses.  Committed:
      checkpoint_nf8.pth        the pickle, written by the reference's classes
      ckpt_nf8_expected.npz     cat(p.flatten() for p in model.parameters()) after the reference's restore_checkpoint
                                (utils.py:7-19) + ema.copy_to (ema.py:53-64)
      ckpt_layout.json          names and shapes of net.named_parameters() at the real width nf = 128 (564 tensors)"""
    import numpy as np
    import torch
    _stub_cifar_env()
    import CIFAR10NaturalInference as R
    from models.ema import ExponentialMovingAverage
    torch.manual_seed(31)
    config = R.configs.get_config()
    config.device = torch.device("cpu")
    full = R.mutils.create_model(config).module
    layout = [(n, list(p.shape)) for n, p in full.named_parameters()]
    (HERE / "ckpt_layout.json").write_text(json.dumps(dict(names=[n for n, _ in layout], shapes=[s for _, s in layout],
                                                              n_param=int(sum(int(np.prod(s)) for _, s in layout))), indent=0))
    del full
    config.model.nf = 8
    model = R.mutils.create_model(config)                                    # DataParallel(NCSNpp)
    ema = ExponentialMovingAverage(model.parameters(), decay=config.model.ema_rate)
    optimizer = torch.optim.Adam(model.parameters(), lr=2e-4)
    with torch.no_grad():                                                    # stand-in for optimiser steps + EMA updates
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
    ema.update(model.parameters())
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
    state = dict(optimizer=optimizer, model=model, ema=ema, step=8)
    # utils.save_checkpoint's body (utils.py:22-29; the module itself imports tensorflow at the top)
    saved_state = {'optimizer': state['optimizer'].state_dict(), 'model': state['model'].state_dict(),
                   'ema': state['ema'].state_dict(), 'step': state['step']}
    torch.save(saved_state, HERE / "checkpoint_nf8.pth")
    # restore_checkpoint's body (utils.py:14-19) + ema.copy_to on a fresh model, as src/CIFAR10NaturalInference.py:258-265 does
    model2 = R.mutils.create_model(config)
    ema2 = ExponentialMovingAverage(model2.parameters(), decay=config.model.ema_rate)
    opt2 = torch.optim.Adam(model2.parameters(), lr=2e-4)
    st = dict(optimizer=opt2, model=model2, ema=ema2, step=0)
    loaded_state = torch.load(HERE / "checkpoint_nf8.pth", map_location="cpu", weights_only=False)
    st['optimizer'].load_state_dict(loaded_state['optimizer'])
    st['model'].load_state_dict(loaded_state['model'], strict=False)
    st['ema'].load_state_dict(loaded_state['ema'])
    st['step'] = loaded_state['step']
    ema2.copy_to(model2.parameters())
    flat = torch.cat([p.detach().flatten() for p in model2.parameters()]).numpy()
    raw = torch.cat([p.detach().flatten() for p in model.parameters()]).numpy()
    assert np.abs(flat - raw).max() > 1e-3                                   # EMA != raw weights: the test can tell them apart
    np.savez_compressed(HERE / "ckpt_nf8_expected.npz", flat=flat, names=np.array([n for n, _ in model2.module.named_parameters()]))
    from naturaldiffusion_amd.ncsnpp import load_score_sde_checkpoint       # the product loader must agree, here and now
    got = load_score_sde_checkpoint(str(HERE / "checkpoint_nf8.pth"), nf=8).numpy()
    assert np.array_equal(got, flat), "load_score_sde_checkpoint differs from restore_checkpoint + ema.copy_to"
    print("ckpt:", len(layout), "parameters at nf=128;", flat.size, "floats at nf=8; loader agrees")


GROUPS = dict(k5=group_k5, ddim_vp=group_ddim_vp, ddpm=group_ddpm, ckpt=group_ckpt, dit=group_dit, loaders=group_loaders, cifar=group_cifar, validate=group_validate, sd3=group_sd3, ncsnpp=group_ncsnpp)

if __name__ == "__main__":
    assert REF.exists(), "the reference is only mounted in the build container"
    if len(sys.argv) > 1:
        GROUPS[sys.argv[1]]()
    else:
        for name in GROUPS:                      # separate processes: the scripts' stubs collide
            subprocess.check_call([sys.executable, __file__, name])
