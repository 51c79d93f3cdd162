"""Pin oracle/ddpm_oracle.py to the reference's own ``DDPM`` nn.Module (deps/score_sde_pytorch/models/ddpm.py:39-181) run on the same
synthetic weights (tests/golden/make_golden.py group ``ddpm``): the 2-res-block network of the checkpoint the reference's docstring names."""
import numpy as np
import pytest
import torch

from oracle import ddpm_oracle as D


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(golden_dir / "ddpm_forward.npz")


@pytest.fixture(scope="module")
def run(fx):
    P = D.make_params(seed=0)
    taps = {}
    y = D.forward(P, torch.from_numpy(fx["x"]), torch.from_numpy(fx["labels"]), taps)
    return P, y, taps


def test_param_inventory(fx, run):
    P, _, _ = run
    assert sum(v.numel() for v in P.values()) == int(fx["n_param"]) == 35218947
    assert len(D.plan()) == int(fx["n_modules"]) == 37
    assert list(P.keys()) == list(fx["names"])                          # the reference's named_parameters() order, name for name


def test_forward_matches_reference_module(fx, run):
    _, y, taps = run
    ref = fx["y"]
    assert np.abs(y.numpy() - ref).max() <= 2e-5 * np.abs(ref).max()
    assert len(taps) == 37
    for k, t in taps.items():
        assert list(t.shape) == list(fx[f"tap{k:02d}_shape"])
        st = fx[f"tap{k:02d}_stats"]
        assert abs(t.mean().item() - st[0]) <= 1e-5 * max(1.0, st[2])
        assert abs(t.std().item() - st[1]) <= 1e-5 * max(1.0, st[2])
        assert np.abs(t.flatten()[:32].numpy() - fx[f"tap{k:02d}_head"]).max() <= 2e-5 * st[2]
    for k in (5, 6, 7, 19, 22):
        full = fx[f"tap{k:02d}_full"]
        assert np.abs(taps[k].numpy() - full).max() <= 2e-5 * np.abs(full).max()


def test_engine_plan_matches_the_oracle_plan():
    """the C++ plan builder's `ddpm` module list and parameter walk == the oracle's (and so the reference's) -- no GPU needed"""
    from naturaldiffusion_amd import ncsnpp
    tab = ncsnpp.module_table(arch="ddpm")
    ref = D.plan()
    assert len(tab) == len(ref) == 37
    for (idx, kind, cin, cout, up, down, res, _), m in zip(tab, ref):
        assert (idx, kind, cin, cout, res) == (m.idx, m.kind, m.cin, m.cout, m.res) and not up and not down
    lay = ncsnpp.param_layout(arch="ddpm")
    assert [(n, tuple(s)) for n, s in lay] == [(n, tuple(s)) for n, s in D.param_shapes().items()]
    flat = ncsnpp.flatten_state_dict(D.make_params(0), arch="ddpm")
    assert flat.numel() == 35218947
    assert abs(D.flops_per_image() / 1e9 - 12.04) < 0.01             # 2 x MAC of the convolutions / linears / attention (NCSN++: 21.69)
