"""CPU checks of the FID Inception-V3 restatement (oracle/inception_oracle.py, PARITY UNPINNED: pytorch_fid / torchvision are not in the
image) and of the host-side parameter plumbing of the HIP engine (naturaldiffusion_amd/inception.py)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from naturaldiffusion_amd import inception as I
from oracle import inception_oracle as O


def test_product_and_oracle_agree_on_the_layer_table():
    assert I.conv_specs() == O.conv_specs() and I.param_layout() == O.param_layout()
    assert len(O.conv_specs()) == 94                                     # BasicConv2d modules on the pool3 path
    assert O.n_params() == 21_820_000                                    # conv filters + 4 BatchNorm vectors each (no AuxLogits / fc)
    assert abs(O.flops_per_image() / 1e9 - 11.42) < 0.01                 # 5.71 GMAC at 299 x 299: the published Inception-V3 figure


def test_engine_plan_walks_the_same_parameters():
    from naturaldiffusion_amd._lib import lib
    h = C.c_void_p()
    assert lib.natinf_inception_create(C.byref(h), 32, 32) == 0
    try:
        assert lib.natinf_inception_param_count(h) == O.n_params()
        assert lib.natinf_inception_workspace_bytes(h, 2) == 2 * lib.natinf_inception_workspace_bytes(h, 1) > 0
    finally:
        lib.natinf_inception_destroy(h)
    assert lib.natinf_inception_create(C.byref(h), 0, 32) != 0


def test_flatten_accepts_both_naming_schemes():
    P = O.make_params(3)
    flat = I.flatten_state_dict(P)
    assert flat.numel() == O.n_params()
    inv = {v: k for k, v in I._FID_BLOCKS.items()}
    fid_named = {}
    for k, v in P.items():
        mod = k.split(".")[0]
        fid_named[".".join([inv[mod]] + k.split(".")[1:])] = v
    fid_named["fc.weight"] = torch.zeros(3)                              # extra keys are ignored
    assert torch.equal(I.flatten_state_dict(fid_named), flat)
    bad = dict(P); del bad["Mixed_7c.branch_pool.bn.running_var"]
    with pytest.raises(KeyError):
        I.flatten_state_dict(bad)


def test_oracle_forward_properties():
    P = O.make_params(0)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(3, 3, 32, 32, generator=g)
    taps = {}
    y = O.forward(P, x, taps)
    assert tuple(y.shape) == (3, 2048) and torch.isfinite(y).all() and (y >= 0).all()        # an average of ReLU outputs
    assert tuple(taps["Mixed_5d"].shape[1:]) == (288, 35, 35) and tuple(taps["Mixed_6e"].shape[1:]) == (768, 17, 17)
    assert tuple(taps["Mixed_7c"].shape[1:]) == (2048, 8, 8)
    # a sample's features do not depend on its batch neighbours (BatchNorm in eval mode)
    assert torch.allclose(O.forward(P, x[1:2]), y[1:2], atol=1e-5, rtol=1e-5)
    # the uint8 / 255 route of the reference (get_activation) is the float route
    u8 = (x * 255).round().to(torch.uint8)
    assert torch.equal(O.forward(P, u8.float() / 255), O.forward(P, (u8.to(torch.float32) / 255)))


def test_fid_average_pool_ignores_the_padding():
    """the FID patch: F.avg_pool2d(..., count_include_pad=False) -- a constant map stays constant at the border"""
    h = torch.full((1, 8, 5, 5), 2.0)
    assert torch.equal(F.avg_pool2d(h, 3, 1, 1, count_include_pad=False), h)
    assert not torch.equal(F.avg_pool2d(h, 3, 1, 1), h)
