"""The one-launch forms of the 16x16 attention block (csrc/attn_blk256.h).  The DEFAULT plan since round 6 is k_attn_blk256_v2 (natinf_set_attn_block(2): q k^T and P V against the
normalised tokens themselves through folded weight matrices -- another arithmetic, bounded below against fp32 and, per module, in tests/test_gpu_ncsnpp.py); plans 1 and 0 are
the four-projection kernels, byte-identical to each other:
k_attn_blk256 (plan 1): the whole 16x16 attention block of NCSN++ (AttnBlockpp, deps/score_sde_pytorch/models/layerspp.py:75-91) as ONE launch -- q stays in
registers, k and V^T are written and re-read through L2 by the same block -- against the two launches it replaces (k_qkv256 + k_attn256<true, 8>, natinf_set_attn_block(0)).
Every output element is the same arithmetic in the same order (the wave's queries are only taken in another order), and since round 6 the GroupNorm partial sums of the
block's output are added up in the two launches' order too (dpp_row_sum_tau: the same addition tree over the wave's 32 queries), so EVERY module and the network's output
are the same bytes under both plans.  The per-module taps against the fp32 oracle (tests/test_gpu_ncsnpp.py) run on the default plan, i.e. on this kernel; the block alone
against an fp32 attention on the SAME input: test_attention_block_alone_against_fp32 below."""
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
        lib.natinf_set_attn_block(-1)
    assert torch.isfinite(outs[1]).all()
    for k in (9, 11, 46):               # the first attention block, the second (its input went through the first one's GroupNorm statistics), the up path's
        assert torch.equal(taps[1][k], taps[0][k]), "module %d: %g" % (k, (taps[1][k] - taps[0][k]).abs().max().item())
    assert torch.equal(outs[1], outs[0]), "network output: %g" % (outs[1] - outs[0]).abs().max().item()


@pytest.mark.parametrize("block", [2, 1, 0])
def test_attention_block_alone_against_fp32(block):
    """AttnBlockpp (layerspp.py:75-91) in fp32 (oracle.ncsnpp_oracle.attn_block: GroupNorm, q / k / v NIN, softmax(q k^T / sqrt C) v, NIN_3, (x + h) / sqrt 2) on the
    engine's OWN module-8 output, against the engine's module 9: the block's error alone, upstream error excluded (round-5 review, item 5: the one-launch block had only
    been compared with the two launches).  Plans 1 and 0: the same figure, they are the same bytes; plan 2 (k_attn_blk256_v2: q k^T and P V against h itself through the
    folded matrices Wq Wk^T and Wv W3 -- another arithmetic, exact in real numbers): the same bound."""
    from naturaldiffusion_amd._lib import lib, check
    from naturaldiffusion_amd.ncsnpp import NCSNppEngine
    from naturaldiffusion_amd.synth import synthetic_flat_params, synthetic_state_dict
    from oracle import ncsnpp_oracle as N
    B = 6
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, 3, 32, 32, generator=g).cuda()
    t = (torch.rand(B, generator=g) * 999).cuda()
    try:
        check(lib.natinf_set_attn_block(block), "knob")
        eng = NCSNppEngine(synthetic_flat_params(0), max_batch=B, keep_activations=True)
        eng(x, t)
        x8, got = eng.tap(8, (B, 256, 16, 16)).cpu(), eng.tap(9, (B, 256, 16, 16)).cpu()
        del eng
    finally:
        lib.natinf_set_attn_block(-1)
    ref = N.attn_block(x8, synthetic_state_dict(0), "all_modules.9.")
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    print("attention block alone (plan %d) vs fp32: max rel %.3e" % (block, err))
    assert torch.isfinite(got).all() and err <= 8e-3, err            # bf16 operands (h, q, k, v, P, O) and a bf16 output: a few 2^-9 steps of the output's range
