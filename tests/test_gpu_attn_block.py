"""k_attn_blk256 (csrc/attn_blk256.h): the whole 16x16 attention block of NCSN++ (AttnBlockpp, deps/score_sde_pytorch/models/layerspp.py:75-91) as ONE launch -- q stays in
registers, k and V^T are written and re-read through L2 by the same block -- against the two launches it replaces (k_qkv256 + k_attn256<true, 8>, natinf_set_attn_block(0)).
Every output element is the same arithmetic in the same order (the wave's queries are only taken in another order), so the FIRST attention block's output (module 9: its
input is computed by the same kernels in both plans) must be the same bytes; behind it the GroupNorm partial sums of that output are added up in another order (last-bit
differences of a statistic), so later modules agree to bf16 rounding flips, not bit for bit.  The per-module taps against the fp32 oracle (tests/test_gpu_ncsnpp.py) run on
the default plan, i.e. on this kernel."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B", [3, 40])
def test_one_launch_attention_block_equals_the_two_launches(B):
    from naturaldiffusion_amd._lib import lib, check
    from naturaldiffusion_amd.ncsnpp import NCSNppEngine
    from naturaldiffusion_amd.synth import synthetic_flat_params
    flat = synthetic_flat_params(0)
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, 3, 32, 32, generator=g).cuda()
    t = (torch.rand(B, generator=g) * 999).cuda()
    outs, taps = {}, {}
    try:
        for on in (1, 0):
            check(lib.natinf_set_attn_block(on), "knob")
            eng = NCSNppEngine(flat, max_batch=B, keep_activations=True)
            outs[on] = eng(x, t).clone()
            taps[on] = {k: eng.tap(k, (B, 256, 16, 16)).clone() for k in (9, 11, 46)}
            torch.cuda.synchronize()
            del eng
    finally:
        lib.natinf_set_attn_block(1)
    assert torch.isfinite(outs[1]).all()
    assert torch.equal(taps[1][9], taps[0][9]), "first attention block: %g" % (taps[1][9] - taps[0][9]).abs().max().item()
    for k in (11, 46):
        e = ((taps[1][k] - taps[0][k]).abs().max() / taps[0][k].abs().max()).item()
        assert e <= 2e-2, (k, e)
    e = ((outs[1] - outs[0]).abs().max() / outs[0].abs().max()).item()
    print("network output, one launch vs two: max rel", e)
    assert e <= 3e-2, e
