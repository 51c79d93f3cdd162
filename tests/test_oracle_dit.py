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


def test_vae_and_mmdit_oracles_have_the_published_sizes():
    """sanity anchors for the two unpinned restatements: parameter counts of the public checkpoints (sd-vae-ft-ema decoder
    49.5 M; SD3-medium transformer 2.03 B without the position table) and output shapes."""
    from oracle import vae_oracle as V, mmdit_oracle as M
    n = sum(int(np.prod(s)) for s in V.param_shapes(4).values())
    assert n == 49490179
    y = V.decode(V.make_params(4, 0), torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(0)))
    assert y.shape == (1, 3, 64, 64) and torch.isfinite(y).all()
    sh = M.param_shapes(24, 24, 4096, 2048)
    assert sum(int(np.prod(s)) for k, s in sh.items() if k != "pos_embed.pos_embed") == 2028328000
