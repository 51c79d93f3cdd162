"""Pin the DiT oracle (oracle/dit_oracle.py) to outputs of the reference's own ``DiT`` class (deps/DiT/models.py)
captured on the same synthetic weights (tests/golden/make_golden.py, group ``dit``)."""
import numpy as np
import pytest
import torch

from oracle import dit_oracle as D


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(golden_dir / "dit_forward.npz")


@pytest.mark.parametrize("tag,depth,hid,heads", [("s64", 2, 128, 2), ("s72", 1, 576, 8)])
def test_small_configs_match_reference_class(fx, tag, depth, hid, heads):
    P = D.make_params(depth, hid, seed=7)
    y = D.forward(P, torch.from_numpy(fx[f"{tag}_x"]), torch.from_numpy(fx[f"{tag}_t"]), torch.from_numpy(fx[f"{tag}_y"]), heads)
    ref = fx[f"{tag}_out"]
    assert y.shape == ref.shape
    assert np.abs(y.numpy() - ref).max() <= 1e-5 * np.abs(ref).max()


def test_xl2_parameter_count(fx):
    n = sum(int(np.prod(s)) for k, s in D.param_shapes(28, 1152).items())
    assert n == int(fx["xl2_nparam"]) == 675129632                  # DiT-XL/2 (models.py:333-334), incl. the frozen pos_embed
