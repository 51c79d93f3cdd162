"""Pin the NCSN++ oracle (oracle/ncsnpp_oracle.py) to activations captured from the
reference's own ``NCSNpp`` nn.Module run on the same synthetic weights
(tests/golden/make_golden.py, group ``ncsnpp``)."""
import numpy as np
import pytest
import torch

from oracle import ncsnpp_oracle as N


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(golden_dir / "ncsnpp_forward.npz")


@pytest.fixture(scope="module")
def run(fx):
    P = N.make_params(seed=0)
    taps = {}
    y = N.forward(P, torch.from_numpy(fx["x"]), torch.from_numpy(fx["labels"]), taps)
    return P, y, taps


def test_param_inventory(fx, run):
    P, _, _ = run
    assert sum(v.numel() for v in P.values()) == int(fx["n_param"]) == 61804419
    assert len(N.plan()) == 55
    assert abs(N.flops_per_image() / 1e9 - 21.69) < 0.01


def test_forward_matches_reference_module(fx, run):
    _, y, taps = run
    ref = fx["y"]
    assert y.shape == ref.shape
    assert np.abs(y.numpy() - ref).max() <= 2e-5 * np.abs(ref).max()
    assert len(taps) == 55
    for k, t in taps.items():
        assert list(t.shape) == list(fx[f"tap{k:02d}_shape"])
        st = fx[f"tap{k:02d}_stats"]
        assert abs(t.mean().item() - st[0]) <= 1e-5 * max(1.0, st[2])
        assert abs(t.std().item() - st[1]) <= 1e-5 * max(1.0, st[2])
        assert np.abs(t.flatten()[:32].numpy() - fx[f"tap{k:02d}_head"]).max() <= 2e-5 * st[2]
    for k in (7, 8, 9, 27, 29, 34, 45):
        full = fx[f"tap{k:02d}_full"]
        assert np.abs(taps[k].numpy() - full).max() <= 2e-5 * np.abs(full).max()
