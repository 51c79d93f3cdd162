"""GPU parity of the `ddpm` plan of the NCSN++ engine (NATINF_NCSNPP_DDPM; reference deps/score_sde_pytorch/models/ddpm.py:39-181 under
configs/vp/ddpm/cifar10_continuous.py -- the checkpoint the reference's docstring names, src/CIFAR10NaturalInference.py:416) against
oracle/ddpm_oracle.py, itself pinned to the reference's DDPM class by tests/golden/ddpm_forward.npz.  Same bars as the NCSN++ tests."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ddpm_oracle as D
from oracle import ni_oracle as O

TOL, TOL_MODULE = 3e-2, 4e-2


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def params():
    return D.make_params(seed=0)


@pytest.fixture(scope="module")
def flat(params):
    from naturaldiffusion_amd.ncsnpp import flatten_state_dict
    return flatten_state_dict(params, arch="ddpm")


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


def test_forward_per_module(dev, params, flat, golden_dir, repo_root):
    from naturaldiffusion_amd.ncsnpp import NCSNppEngine
    fx = np.load(golden_dir / "ddpm_forward.npz")
    x, labels = torch.from_numpy(fx["x"]), torch.from_numpy(fx["labels"])
    taps = {}
    y_ref = D.forward(params, x, labels, taps)
    assert np.abs(y_ref.numpy() - fx["y"]).max() <= 2e-5 * np.abs(fx["y"]).max()       # oracle == the reference's DDPM module
    eng = NCSNppEngine(flat, max_batch=2, device=dev, keep_activations=True, arch="ddpm")
    y = eng(x.to(dev), labels.to(dev))
    torch.cuda.synchronize()
    report = {}
    for k in range(2, 35):
        report[f"tap{k:02d}"] = _rel(eng.tap(k, tuple(taps[k].shape)).cpu(), taps[k])
    report["y"] = _rel(y.cpu(), y_ref)
    os.makedirs(repo_root / "gpurun_out", exist_ok=True)
    (repo_root / "gpurun_out" / "ddpm_tap_errors.json").write_text(json.dumps(report, indent=1))
    print("ddpm per-module max-rel errors:", json.dumps(report))
    assert torch.isfinite(y).all()
    first_bad = next((k for k in range(2, 35) if report[f"tap{k:02d}"] > TOL_MODULE), None)
    assert first_bad is None, f"module {first_bad} first exceeds {TOL_MODULE}: {report[f'tap{first_bad:02d}']:.3e}"
    assert report["y"] <= TOL, report["y"]


def test_batch_512_and_batch_independence(dev, flat, golden_dir):
    """the production plan at the benchmark batch: the golden samples in slots 0-1 and 510-511 match the reference module's output and
    each other bit for bit; the fused GroupNorm + convolution kernel and the fused head carry this network too"""
    from naturaldiffusion_amd.ncsnpp import NCSNppEngine
    fx = np.load(golden_dir / "ddpm_forward.npz")
    gx, gl = torch.from_numpy(fx["x"]), torch.from_numpy(fx["labels"])
    g = torch.Generator().manual_seed(13)
    x = torch.randn(512, 3, 32, 32, generator=g); labels = torch.rand(512, generator=g) * 999
    for s in (0, 510):
        x[s:s + 2] = gx; labels[s:s + 2] = gl
    eng = NCSNppEngine(flat, max_batch=512, device=dev, arch="ddpm")
    kinds = [r[6].split("/")[0] for r in eng.describe_gemms(512)]
    assert kinds.count("conv_gn") >= 20 and kinds.count("head_conv") == 1, {k: kinds.count(k) for k in set(kinds)}
    y = eng(x.to(dev), labels.to(dev))
    torch.cuda.synchronize()
    ref = torch.from_numpy(fx["y"])
    head, tail = y[0:2].cpu(), y[510:512].cpu()
    assert torch.isfinite(y).all() and _rel(head, ref) <= TOL and torch.equal(head, tail)
    y2 = NCSNppEngine(flat, max_batch=2, device=dev, arch="ddpm")(gx.to(dev), gl.to(dev)).cpu()
    assert _rel(head, y2) < 2e-2


def test_ni_end_to_end_with_the_ddpm_engine(dev, params, flat, repo_root):
    """BASELINE config 1's coefficient file through the `ddpm` denoiser: HIP engine + ni_step vs the all-CPU oracle path"""
    from naturaldiffusion_amd.ncsnpp import NCSNppEngine
    from naturaldiffusion_amd.CIFAR10NaturalInference import natural_inference
    C, B, node = O.load_coeff_npz(repo_root / "weights/step_5_weight_00.npz")
    noise = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(888))
    ref = O.cifar_ni_trajectory(D.model_fn_from_params(params), noise, C, B, node)[-1]
    eng = NCSNppEngine(flat, max_batch=2, device=dev, arch="ddpm")
    got = natural_inference(eng, noise.to(dev), repo_root / "weights/step_5_weight_00.npz").cpu()
    assert _rel(got, ref) <= 5e-2
