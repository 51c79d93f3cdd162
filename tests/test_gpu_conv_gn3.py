"""k_conv_gn3 (csrc/conv_gn3.h: the fused GroupNorm-apply + SiLU + 3x3 convolution on ONE wave per SIMD, 128 x 128 wave tiles, a K loop written slot by
slot) on its own, selected by natinf_set_conv_gn_w128: (a) against plain PyTorch of the same op -- conv2d(silu(x * scale + shift), w, padding=1) + 1x1
shortcut + bias + residual, scaled (reference arithmetic: ResnetBlockBigGANpp.forward, deps/score_sde_pytorch/models/layerspp.py:242-274), tolerance
1e-2 of max |ref| as for k_conv_gn2; (b) against k_conv_gn2 BIT FOR BIT: same K order, same normalisation arithmetic, one accumulation chain per output
element; (c) the GroupNorm partial sums of the output per 512- / 256-pixel tile against the reference's."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
LOG2E = 1.4426950408889634


def _pack(w, w1):
    N, C = w.shape[:2]
    p = w.reshape(N, C // 64, 64, 9).permute(0, 1, 3, 2).reshape(N, 9 * C)
    return torch.cat([p, w1], dim=1).contiguous() if w1 is not None else p.contiguous()


def _case(res, B, cin, N, c1, resid, up_flags, seed):
    g = torch.Generator().manual_seed(seed)
    bf = lambda t: t.bfloat16().float()
    h2 = res // 2
    xr = h2 if up_flags & 1 else res
    ar = h2 if up_flags & 2 else res
    x = bf(torch.randn(B, xr, xr, cin, generator=g))
    scale = torch.rand(B, cin, generator=g) * 1.5 + 0.25
    shift = torch.randn(B, cin, generator=g) * 0.5
    w = bf(torch.randn(N, cin, 3, 3, generator=g) / np.sqrt(9 * cin))
    w1 = bf(torch.randn(N, c1, generator=g) / np.sqrt(c1)) if c1 else None
    a1 = bf(torch.randn(B, ar, ar, c1, generator=g)) if c1 else None
    bias = torch.randn(N, generator=g) * 0.1
    r = bf(torch.randn(B * res * res, N, generator=g)) if resid else None
    up = lambda t: F.interpolate(t.permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1)
    xf = up(x) if up_flags & 1 else x
    h = bf(F.silu(xf * scale[:, None, None, :] + shift[:, None, None, :]))
    ref = F.conv2d(h.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1).reshape(B * res * res, N)
    if c1:
        ref = ref + (up(a1) if up_flags & 2 else a1).reshape(-1, c1).double() @ w1.double().t()
    ref = ref + bias.double()
    if resid:
        ref = ref + r.double()
    return dict(x=x, scale=scale, shift=shift, w=w, w1=w1, a1=a1, bias=bias, r=r), (ref * 0.70710678).float()


def _run(res, B, cin, N, c1, t, up_flags, mask, rows, parts):
    from naturaldiffusion_amd._lib import lib, check, ptr, stream_ptr
    dev = "cuda"
    M = B * res * res
    wd = _pack(t["w"] * (-1.0 / LOG2E), t["w1"]).bfloat16().to(dev)
    xd = t["x"].bfloat16().to(dev).contiguous()
    scd, shd, bd = (t["scale"] * -LOG2E).to(dev), (t["shift"] * -LOG2E).to(dev), t["bias"].to(dev)
    a1d = t["a1"].bfloat16().to(dev).contiguous() if c1 else None
    rd = t["r"].bfloat16().to(dev) if t["r"] is not None else None
    wf = torch.zeros_like(wd)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    part = torch.zeros(M // rows, N // 4, 2, device=dev) if parts else None
    check(lib.natinf_set_conv_gn_w128(mask), "knob")
    for sh in range(3):
        check(lib.natinf_set_conv_gn_w128_min_k(sh, 0), "min_k")               # every K: the library's defaults keep short-K launches on k_conv_gn2
    check(lib.natinf_debug_conv_gn_up(up_flags), "up")
    try:
        check(lib.natinf_debug_conv_gn(res, B, N, cin, c1, ptr(xd), ptr(scd), ptr(shd), ptr(wd), ptr(wf), ptr(a1d), ptr(bd), ptr(rd), 0.70710678, ptr(out),
                                       ptr(part), 1, stream_ptr()), "conv_gn")
        torch.cuda.synchronize()
    finally:
        lib.natinf_set_conv_gn_w128(_DEFAULT_MASK)
        for sh, k in enumerate(_DEFAULT_MIN_K):
            lib.natinf_set_conv_gn_w128_min_k(sh, k)
        lib.natinf_debug_conv_gn_up(0)
    return out.cpu(), (part.cpu() if parts else None)


_DEFAULT_MASK, _DEFAULT_MIN_K = 7, (2304, 0, 2304)          # (csrc/ncsnpp.hip: g_cg3, g_cg3_min_k)


@pytest.mark.parametrize("res,B,cin,N,c1,resid,parts,up", [
    (32, 2, 128, 128, 0, False, True, 0),        # Conv_0 of a level-0 block: 512 x 128 tiles, four half-chunks
    (32, 1, 128, 128, 0, True, True, 0),         # Conv_1 with the identity residual, a single image (two tiles)
    (32, 3, 256, 128, 256, False, True, 0),      # up-path Conv_1 with the 1x1 shortcut segment (eight shortcut steps)
    (32, 2, 384, 128, 384, False, False, 0),     # 384 channels: 12 half-chunks + 12 shortcut steps
    (32, 2, 128, 128, 128, False, True, 0),      # four shortcut steps
    (32, 2, 64, 128, 64, True, True, 0),         # the shortest K: two half-chunks (the loop body never runs), two shortcut steps
    (32, 2, 256, 256, 0, False, True, 0),        # N = 256 at 32x32: 256 x 256 tiles
    (32, 1, 256, 256, 256, True, True, 0),       # ... with the shortcut segment and a residual
    (16, 4, 256, 256, 0, True, True, 0),         # 16x16: one image per 256 x 256 tile
    (16, 3, 512, 256, 512, False, True, 0),      # widest K: 16 half-chunks + 16 shortcut steps, an odd batch
    (16, 3, 128, 256, 0, False, False, 0),
    (32, 2, 256, 128, 0, False, True, 1),        # up-sampled patch fetch (Conv_0 of the 16 -> 32 block)
    (32, 2, 128, 128, 256, False, True, 3),      # up-sampled patch AND shortcut fetch
    (32, 1, 256, 256, 256, False, True, 3),      # ... at N = 256
    (16, 3, 256, 256, 256, False, True, 3),      # the 8 -> 16 block
])
def test_conv_gn3_matches_torch_and_conv_gn2_bit_for_bit(res, B, cin, N, c1, resid, parts, up):
    t, ref = _case(res, B, cin, N, c1, resid, up, res * 1000 + cin + N + c1 + 17 * up)
    M = B * res * res
    rows3 = 512 if (res == 32 and N % 256) else 256
    rows2 = 128 if N % 256 == 0 else 256
    got3, part3 = _run(res, B, cin, N, c1, t, up, 7, rows3, parts)
    got2, _ = _run(res, B, cin, N, c1, t, up, 0, rows2, parts)
    assert torch.isfinite(got3.float()).all()
    err = ((got3.float() - ref).abs().max() / ref.abs().max()).item()
    assert err <= 1e-2, err
    assert torch.equal(got3.view(torch.int16), got2.view(torch.int16)), "k_conv_gn3 and k_conv_gn2 differ: max |d| %g" % (got3.float() - got2.float()).abs().max().item()
    if parts:
        want = torch.stack([ref.reshape(M // rows3, rows3, N // 4, 4).sum(dim=(1, 3)), (ref ** 2).reshape(M // rows3, rows3, N // 4, 4).sum(dim=(1, 3))], dim=-1)
        assert ((part3 - want).abs().max() / want.abs().max()).item() <= 5e-3


def test_conv_gn3_is_reproducible():
    """Two runs give the same bytes."""
    res, B, cin, N, c1 = 32, 2, 128, 128, 128
    t, _ = _case(res, B, cin, N, c1, True, 0, 5)
    a, _ = _run(res, B, cin, N, c1, t, 0, 7, 512, False)
    b, _ = _run(res, B, cin, N, c1, t, 0, 7, 512, False)
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))


def test_the_default_rule_sends_long_k_launches_to_conv_gn3():
    """natinf_set_conv_gn_w128_min_k defaults (2304 / 0 / 2304): the GroupNorm partial rows tell which kernel ran -- 512 / 256 pixels per row for k_conv_gn3,
    256 / 128 for k_conv_gn2."""
    for res, B, cin, N, c1, rows in ((32, 1, 128, 128, 0, 256), (32, 1, 256, 128, 0, 512), (32, 1, 128, 256, 0, 256), (16, 2, 128, 256, 0, 128), (16, 2, 256, 256, 0, 256)):
        t, ref = _case(res, B, cin, N, c1, False, 0, 77 + res + cin + c1)
        from naturaldiffusion_amd._lib import lib, check, ptr, stream_ptr
        dev, M = "cuda", B * res * res
        wd = _pack(t["w"] * (-1.0 / LOG2E), t["w1"]).bfloat16().to(dev)
        xd = t["x"].bfloat16().to(dev).contiguous()
        scd, shd, bd = (t["scale"] * -LOG2E).to(dev), (t["shift"] * -LOG2E).to(dev), t["bias"].to(dev)
        a1d = t["a1"].bfloat16().to(dev).contiguous() if c1 else None
        wf = torch.zeros_like(wd)
        out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        part = torch.full((M // 128, N // 4, 2), -1.0, device=dev)              # (room for the smallest rows; unwritten rows keep -1)
        check(lib.natinf_debug_conv_gn(res, B, N, cin, c1, ptr(xd), ptr(scd), ptr(shd), ptr(wd), ptr(wf), ptr(a1d), ptr(bd), None, 0.70710678, ptr(out), ptr(part), 1,
                                       stream_ptr()), "conv_gn")
        torch.cuda.synchronize()
        written = int((part[:, 0, 1] >= 0).sum().item())                        # (a sum of squares is never negative)
        assert written == M // rows, (res, cin, N, c1, written, M // rows)
        assert ((out.float().cpu() - ref).abs().max() / ref.abs().max()).item() <= 1e-2
