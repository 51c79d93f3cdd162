"""Checkpoint boundary (SURVEY section 8b, row B-iii): the score_sde pickle format and the parameter order.

Fixtures (tests/golden/make_golden.py, group ``ckpt``; written in the build container by the reference's own classes):
  ckpt_layout.json       names / shapes of ``NCSNpp.named_parameters()`` at the real width (nf = 128): the EMA
                         ``shadow_params`` list has exactly this order (ema.py:28-29)
  checkpoint_nf8.pth     utils.save_checkpoint's dict written through DataParallel.state_dict() and
                         ExponentialMovingAverage.state_dict() at width nf = 8
  ckpt_nf8_expected.npz  the parameters after the reference's restore_checkpoint + ema.copy_to on that file
"""
import json

import numpy as np
import pytest
import torch


def test_parameter_order_is_the_reference_modules(golden_dir):
    from naturaldiffusion_amd.ncsnpp import param_layout
    from naturaldiffusion_amd._lib import lib
    fx = json.loads((golden_dir / "ckpt_layout.json").read_text())
    layout = param_layout()
    assert len(layout) == len(fx["names"]) == 564
    assert [n for n, _ in layout] == fx["names"]
    assert [list(s) for _, s in layout] == fx["shapes"]
    assert fx["n_param"] == lib.natinf_ncsnpp_param_count() == 61804419


def test_reference_written_checkpoint_loads_to_the_ema_weights(golden_dir):
    from naturaldiffusion_amd.ncsnpp import load_score_sde_checkpoint, param_layout
    fx = np.load(golden_dir / "ckpt_nf8_expected.npz")
    got = load_score_sde_checkpoint(str(golden_dir / "checkpoint_nf8.pth"), nf=8)
    assert got.dtype == torch.float32 and got.numel() == fx["flat"].size
    assert np.array_equal(got.numpy(), fx["flat"])                       # == restore_checkpoint + ema.copy_to, bit for bit
    assert [n for n, _ in param_layout(8)] == [str(n) for n in fx["names"]]
    # the file really is the reference's format, and the raw (non-EMA) weights are NOT what comes out
    state = torch.load(golden_dir / "checkpoint_nf8.pth", map_location="cpu", weights_only=False)
    assert sorted(state.keys()) == ["ema", "model", "optimizer", "step"]
    assert sorted(state["ema"].keys()) == ["decay", "num_updates", "shadow_params"]
    assert all(k.startswith("module.") for k in state["model"])
    assert "module.sigmas" in state["model"]                             # a buffer: in the state dict, not in the EMA list
    raw = torch.cat([state["model"]["module." + n].flatten() for n, _ in param_layout(8)])
    assert not torch.equal(raw, got)


def test_wrong_width_or_truncated_ema_is_an_error(golden_dir, tmp_path):
    from naturaldiffusion_amd.ncsnpp import load_score_sde_checkpoint
    with pytest.raises(ValueError):
        load_score_sde_checkpoint(str(golden_dir / "checkpoint_nf8.pth"))            # nf = 128 layout vs an nf = 8 file
    state = torch.load(golden_dir / "checkpoint_nf8.pth", map_location="cpu", weights_only=False)
    state["ema"]["shadow_params"] = state["ema"]["shadow_params"][:-1]
    torch.save(state, tmp_path / "short.pth")
    with pytest.raises(ValueError):
        load_score_sde_checkpoint(str(tmp_path / "short.pth"), nf=8)
    torch.save({"model": {}}, tmp_path / "other.pth")
    with pytest.raises(KeyError):
        load_score_sde_checkpoint(str(tmp_path / "other.pth"), nf=8)
