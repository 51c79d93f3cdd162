"""k_conv_gn2 / k_conv_gn (csrc/conv_gn2.h: weights streamed through registers; csrc/conv_gn.h: through an LDS ring) on their own: the 3x3 convolution with GroupNorm-apply + SiLU fused into its operand path, against
plain PyTorch fp32 of the same op -- conv2d(silu(x * scale + shift), w, padding=1) + 1x1 shortcut + bias + residual, scaled
(reference arithmetic: ResnetBlockBigGANpp.forward, deps/score_sde_pytorch/models/layerspp.py:242-274).  Inputs are made
bf16-representable and the activated operand is rounded to bf16 in the reference too, so what remains is fp32 accumulation order
and the bf16 rounding of the output: tolerance 1e-2 of max |ref| (observed ~4e-3)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _pack(w, w1):
    """[N][C][3][3] (+ [N][C1]) -> the engine's K order: ((c / 64) * 9 + tap) * 64 + c % 64, then the shortcut columns"""
    N, C = w.shape[:2]
    p = w.reshape(N, C // 64, 64, 9).permute(0, 1, 3, 2).reshape(N, 9 * C)
    return torch.cat([p, w1], dim=1).contiguous() if w1 is not None else p.contiguous()


@pytest.mark.parametrize("res,B,cin,N,c1,resid,parts", [
    (32, 2, 128, 128, 0, False, True),        # Conv_0 of a level-0 block (+ GroupNorm partials of the output)
    (32, 3, 256, 128, 256, False, True),      # up-path Conv_1 with the 1x1 shortcut segment
    (32, 1, 128, 128, 0, True, True),         # Conv_1 with the identity residual
    (16, 4, 256, 256, 0, True, True),         # 16x16 level, N = 256: the 128 x 256 tile (and, forced, two 256 x 128 tiles)
    (16, 2, 512, 256, 512, False, True),      # widest K: 144 + 16 K-tiles
    (16, 3, 128, 256, 0, False, False),
    (32, 2, 384, 128, 384, False, False),     # 384 channels: 12 half-chunks
    (32, 2, 256, 256, 0, False, True),        # N = 256 at 32x32 (the 16 -> 32 up-sampling block): the 128 x 256 tile (and, forced, two 256 x 128 tiles)
    (32, 1, 256, 256, 256, True, True),       # ... with the shortcut segment and a residual
    (8, 4, 256, 256, 0, False, True),         # 8x8 level (round 3): one whole image per 64-pixel tile, GroupNorm partials per sample
    (8, 3, 512, 256, 512, False, True),       # + the 1x1 shortcut segment
    (8, 6, 256, 256, 0, True, True),          # identity residual
    (8, 1, 256, 256, 256, False, False),      # a single image
    (4, 8, 256, 256, 0, False, True),         # 4x4 level: FOUR whole images per 64-pixel tile, GroupNorm partials per sample
    (4, 5, 512, 256, 512, False, True),       # a batch that is no multiple of four (the last tile holds ONE image) + the 1x1 shortcut segment
    (4, 7, 256, 256, 0, True, True),          # identity residual; three images in the last tile
    (4, 2, 256, 256, 256, False, False),      # half a tile
])
def test_conv_gn_matches_torch(res, B, cin, N, c1, resid, parts):
    from naturaldiffusion_amd._lib import lib, check, ptr, stream_ptr
    g = torch.Generator().manual_seed(res * 1000 + cin + N + c1)
    bf = lambda t: t.bfloat16().float()
    x = bf(torch.randn(B, res, res, cin, generator=g))
    scale = torch.rand(B, cin, generator=g) * 1.5 + 0.25
    shift = torch.randn(B, cin, generator=g) * 0.5
    w = bf(torch.randn(N, cin, 3, 3, generator=g) / np.sqrt(9 * cin))
    w1 = bf(torch.randn(N, c1, generator=g) / np.sqrt(c1)) if c1 else None
    a1 = bf(torch.randn(B, res, res, c1, generator=g)) if c1 else None
    bias = torch.randn(N, generator=g) * 0.1
    r = bf(torch.randn(B * res * res, N, generator=g)) if resid else None
    out_scale = 0.70710678
    # reference
    h = bf(F.silu(x * scale[:, None, None, :] + shift[:, None, None, :]))
    ref = F.conv2d(h.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1).reshape(B * res * res, N)
    if c1:
        ref = ref + a1.reshape(-1, c1).double() @ w1.double().t()
    ref = ref + bias.double()
    if resid:
        ref = ref + r.double()
    ref = (ref * out_scale).float()
    # kernel
    dev = "cuda"
    # the kernel's folded operand form: GroupNorm scale / shift times -log2(e), 3x3 weights times -ln 2 (GemmArgs::gn_folded)
    LOG2E = 1.4426950408889634
    wp = _pack(w * (-1.0 / LOG2E), w1)
    xd, wd = x.bfloat16().to(dev).contiguous(), wp.bfloat16().to(dev)
    out = torch.empty(B * res * res, N, dtype=torch.bfloat16, device=dev)
    M = B * res * res
    scd, shd, bd = (scale * -LOG2E).to(dev), (shift * -LOG2E).to(dev), bias.to(dev)           # (named: a temporary could be recycled before the launch runs)
    a1d = a1.bfloat16().to(dev).contiguous() if c1 else None
    rd = r.bfloat16().to(dev) if resid else None
    wide = res in (16, 32) and N % 256 == 0                # 128-pixel x 256-channel tiles (default) -- the 256 x 128 ones are tested as well
    wf = torch.zeros_like(wd)                              # receives the fragment-major copy of the weights (k_conv_gn2)
    lib.natinf_set_conv_gn_w128(0)                      # this file is about k_conv_gn2 (k_conv_gn3, which takes the long-K launches by default: tests/test_gpu_conv_gn3.py)
    try:
        assert lib.natinf_set_conv_gn_regw(0) != 0          # k_conv_gn (weights through an LDS ring): -DNATINF_DEV builds only
        assert lib.natinf_set_conv_gn8_tile(0) != 0         # the two-images-per-tile form of the 8x8 level: -DNATINF_DEV builds only
        for use_wide, regw in (((1, 1), (0, 1)) if wide else ((1, 1),)):
            lib.natinf_set_conv_gn_wide(3 if use_wide else 0)
            rows = res * res if res <= 8 else (128 if (wide and use_wide) else 256)      # (8x8 / 4x4: one partial row per SAMPLE)
            part = torch.zeros(M // rows, N // 4, 2, device=dev) if parts else None
            out.zero_()
            try:
                check(lib.natinf_debug_conv_gn(res, B, N, cin, c1, ptr(xd), ptr(scd), ptr(shd), ptr(wd), ptr(wf) if regw else None, ptr(a1d), ptr(bd),
                                               ptr(rd), out_scale, ptr(out), ptr(part), 1, stream_ptr()), "conv_gn")
                torch.cuda.synchronize()
            finally:
                lib.natinf_set_conv_gn_wide(3)
            got = out.float().cpu()
            assert torch.isfinite(got).all()
            err = ((got - ref).abs().max() / ref.abs().max()).item()
            assert err <= 1e-2, (err, use_wide, regw)
            if parts:                                           # (sum, sum of squares) per tile and 4-channel quad, of the fp32 results
                want = torch.stack([ref.reshape(M // rows, rows, N // 4, 4).sum(dim=(1, 3)), (ref ** 2).reshape(M // rows, rows, N // 4, 4).sum(dim=(1, 3))], dim=-1)
                assert ((part.cpu() - want).abs().max() / want.abs().max()).item() <= 5e-3
    finally:
        lib.natinf_set_conv_gn_w128(7)                  # the library's default, restored on the failure path too (round-5 advisor note)

def test_ragged_channel_counts_are_refused_in_the_shipped_build():
    """k_conv_gn2 needs whole 128- (or 256-) channel tiles; the LDS-ring kernel that took ragged N is a -DNATINF_DEV kernel now."""
    from naturaldiffusion_amd._lib import lib, ptr, stream_ptr
    dev = "cuda"
    x = torch.zeros(1, 16, 16, 64, dtype=torch.bfloat16, device=dev); sc = torch.ones(1, 64, device=dev); sh = torch.zeros(1, 64, device=dev)
    w = torch.zeros(40, 9 * 64, dtype=torch.bfloat16, device=dev); out = torch.zeros(256, 40, dtype=torch.bfloat16, device=dev)
    assert lib.natinf_debug_conv_gn(16, 1, 40, 64, 0, ptr(x), ptr(sc), ptr(sh), ptr(w), ptr(torch.zeros_like(w)), None, None, None, 1.0, ptr(out), None, 1,
                                    stream_ptr()) == -1


@pytest.mark.parametrize("res,B,cin,N,c1,flags", [
    (32, 2, 256, 128, 0, 1),        # Conv_0 of the 16 -> 32 up-sampling block: the patch is fetched from the 16x16 tensor
    (32, 2, 256, 256, 0, 1),        # ... at its real width (N = 256: the 128 x 256 tile)
    (32, 1, 256, 256, 256, 3),      # Conv_1 + shortcut of that block at N = 256
    (32, 2, 128, 128, 256, 3),      # Conv_1 + shortcut of that block: the 1x1 operand is fetched up-sampled too
    (16, 3, 256, 256, 256, 3),      # the 8 -> 16 block on the 128 x 256 tile
    (16, 2, 256, 256, 0, 1),
])
def test_up_sampling_fetch_paths_match_torch(res, B, cin, N, c1, flags):
    """GemmArgs::a0_up / a1_up (round-2 advisor note: reachable only through the whole network until now): the kernel reads a half-
    resolution raw tensor and up-samples it (nearest 2x) in its patch / shortcut fetches -- against F.interpolate(mode='nearest')."""
    from naturaldiffusion_amd._lib import lib, check, ptr, stream_ptr
    g = torch.Generator().manual_seed(res * 7 + cin + N + c1 + flags)
    bf = lambda t: t.bfloat16().float()
    h2 = res // 2
    x = bf(torch.randn(B, h2, h2, cin, generator=g))
    scale = torch.rand(B, cin, generator=g) * 1.5 + 0.25
    shift = torch.randn(B, cin, generator=g) * 0.5
    w = bf(torch.randn(N, cin, 3, 3, generator=g) / np.sqrt(9 * cin))
    w1 = bf(torch.randn(N, c1, generator=g) / np.sqrt(c1)) if c1 else None
    a1_up = bool(flags & 2)
    a1 = bf(torch.randn(B, h2 if a1_up else res, h2 if a1_up else res, c1, generator=g)) if c1 else None
    bias = torch.randn(N, generator=g) * 0.1
    up = lambda t: F.interpolate(t.permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1)
    hfull = bf(F.silu(up(x) * scale[:, None, None, :] + shift[:, None, None, :]))
    ref = F.conv2d(hfull.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1).reshape(B * res * res, N)
    if c1:
        ref = ref + (up(a1) if a1_up else a1).reshape(-1, c1).double() @ w1.double().t()
    ref = (ref + bias.double()).float()
    dev = "cuda"
    LOG2E = 1.4426950408889634
    wd = _pack(w * (-1.0 / LOG2E), w1).bfloat16().to(dev)
    xd = x.bfloat16().to(dev).contiguous()
    scd, shd, bd = (scale * -LOG2E).to(dev), (shift * -LOG2E).to(dev), bias.to(dev)
    a1d = a1.bfloat16().to(dev).contiguous() if c1 else None
    out = torch.zeros(B * res * res, N, dtype=torch.bfloat16, device=dev)
    wf = torch.zeros_like(wd)
    check(lib.natinf_debug_conv_gn_up(flags), "up")
    try:
        check(lib.natinf_debug_conv_gn(res, B, N, cin, c1, ptr(xd), ptr(scd), ptr(shd), ptr(wd), ptr(wf), ptr(a1d), ptr(bd), None, 1.0, ptr(out), None, 1,
                                       stream_ptr()), "conv_gn")
        torch.cuda.synchronize()
    finally:
        lib.natinf_debug_conv_gn_up(0)
    got = out.float().cpu()
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    assert torch.isfinite(got).all() and err <= 1e-2, err


def test_fragment_major_weight_copy_layout():
    """k_pack_frag (conv_gn2.h): [N/16][K steps][64 lanes][8] with lane l of K step kt holding row r = l & 15 of its n-tile (= output channel 32 (nt >> 1) + 8 (r >> 2) + 4 (nt & 1) + (r & 3)), columns col(kt) + 8 (l >> 4) .. + 7,
    col(kt) = ((hc >> 1) * 9 + tap) * 64 + (hc & 1) * 32 for kt = hc * 9 + tap, then the shortcut columns -- checked element by element against
    a numpy gather of the packed matrix (the copy is what k_conv_gn2 multiplies with)."""
    from naturaldiffusion_amd._lib import lib, check, ptr, stream_ptr
    res, B, cin, N, c1 = 16, 1, 128, 256, 64
    g = torch.Generator().manual_seed(5)
    w = torch.randn(N, 9 * cin + c1, generator=g).bfloat16()
    dev = "cuda"
    x = torch.zeros(B, res, res, cin, dtype=torch.bfloat16, device=dev); a1 = torch.zeros(B * res * res, c1, dtype=torch.bfloat16, device=dev)
    sc = torch.zeros(B, cin, device=dev); sh = torch.zeros(B, cin, device=dev); out = torch.empty(B * res * res, N, dtype=torch.bfloat16, device=dev)
    wd = w.to(dev); wf = torch.zeros_like(wd)
    check(lib.natinf_debug_conv_gn(res, B, N, cin, c1, ptr(x), ptr(sc), ptr(sh), ptr(wd), ptr(wf), ptr(a1), None, None, 1.0, ptr(out), None, 1, stream_ptr()), "conv_gn")
    torch.cuda.synchronize()
    nk, NT = 9 * (cin // 32), 9 * (cin // 32) + c1 // 32
    col = np.array([((kt // 9 >> 1) * 9 + kt % 9) * 64 + (kt // 9 & 1) * 32 if kt < nk else 9 * cin + (kt - nk) * 32 for kt in range(NT)])
    lane = np.arange(64)
    nt, rr = np.arange(N // 16)[:, None, None, None], (lane & 15)[None, None, :, None]
    rows = 32 * (nt >> 1) + 8 * (rr >> 2) + 4 * (nt & 1) + (rr & 3)                # the rows of an n-tile pair are interleaved (eight consecutive channels per accumulator lane)
    cols = col[None, :, None, None] + ((lane >> 4) * 8)[None, None, :, None] + np.arange(8)[None, None, None, :]      # [1][NT][64][8]
    want = w.view(torch.int16).numpy()[np.broadcast_to(rows, (N // 16, NT, 64, 8)), np.broadcast_to(cols, (N // 16, NT, 64, 8))]
    got = wf.cpu().view(torch.int16).numpy().reshape(N // 16, NT, 64, 8)
    assert np.array_equal(got, want)


def test_conv_gn_argument_errors():
    from naturaldiffusion_amd._lib import lib
    d = 4096
    assert lib.natinf_debug_conv_gn(8, 1, 128, 128, 0, d, d, d, d, None, None, None, None, 1.0, d, None, 1, None) == -1      # resolution
    assert lib.natinf_debug_conv_gn(32, 1, 128, 96, 0, d, d, d, d, None, None, None, None, 1.0, d, None, 1, None) == -1      # cin % 64
    assert lib.natinf_debug_conv_gn(32, 1, 128, 128, 64, d, d, d, d, None, None, None, None, 1.0, d, None, 1, None) == -1    # c1 without a1
