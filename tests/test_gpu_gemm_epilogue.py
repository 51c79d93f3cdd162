"""The GEMM epilogue specializations (csrc/gemm_dma.h: packed bf16 / direct fp32 / fp32 slab) one by one, on plain GEMMs with
ragged M and N, against an fp32 torch reference of the same fused terms -- and the packed path against the general one."""
import ctypes as C
import pytest
import torch

pytestmark = pytest.mark.gpu

# the shipped tile variants (ncsnpp.hip variant_shipped): 256x256 / 512x128 with one issuing wave per SIMD, 128x128, the two rings
V_DMA256P, V_DMA128P, V_RING256W4, V_RING64, V_DMA512 = 26, 17, 9, 8, 27
V_W128 = 29        # 256x256x64, one wave per SIMD with a 128x128 wave tile (gemm_w128.h): two 256x128 half-tile epilogues


def _run(variant, M, N, K, lrs, terms, act=0, c_f32=False, fp32_slab=False, scale=1.0, seed=0, splitk=0):
    from naturaldiffusion_amd._lib import lib, check, ptr, stream_ptr
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16(); b = (torch.randn(N, K, device="cuda", generator=g) * 0.1).bfloat16()
    ns = (M + (1 << lrs) - 1) >> lrs
    t = {"bias_n": torch.randn(N, device="cuda", generator=g), "bias_m": torch.randn(M, device="cuda", generator=g),
         "rowvec": torch.randn(ns, N, device="cuda", generator=g), "gate": torch.randn(ns, N, device="cuda", generator=g),
         "resid": torch.randn(M, N, device="cuda", generator=g).bfloat16(), "resid_f32": torch.randn(M, N, device="cuda", generator=g)}
    has = lambda k: k in terms
    c = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32 if c_f32 else torch.bfloat16)
    part = torch.zeros(((M + 15) // 16) * (N // 4) * 2, device="cuda") if has("gn") else None
    ws = torch.full((splitk * M * N,), float("nan"), device="cuda") if splitk else None
    check(lib.natinf_debug_set_splitk_workspace(ptr(ws) if splitk else None, splitk), "set_splitk_workspace")
    bm = C.c_int(0)
    check(lib.natinf_debug_gemm_fused(variant, M, N, K, ptr(a), ptr(b), ptr(t["bias_n"]) if has("bias_n") else None,
                                      ptr(t["bias_m"]) if has("bias_m") else None, ptr(t["rowvec"]) if has("rowvec") else None,
                                      ptr(t["gate"]) if has("gate") else None, lrs, ptr(t["resid"]) if has("resid") else None,
                                      ptr(t["resid_f32"]) if has("resid_f32") else None, scale, act, ptr(c), int(c_f32),
                                      ptr(part) if part is not None else None, C.byref(bm), int(fp32_slab), stream_ptr()), "debug_gemm_fused")
    torch.cuda.synchronize()
    check(lib.natinf_debug_set_splitk_workspace(None, 0), "set_splitk_workspace")
    v = a.float() @ b.float().t()
    rows = torch.arange(M, device="cuda") >> lrs
    if has("bias_n"): v = v + t["bias_n"]
    if has("bias_m"): v = v + t["bias_m"][:, None]
    if has("rowvec"): v = v + t["rowvec"][rows]
    if has("gate"): v = v * t["gate"][rows]
    if has("resid"): v = v + t["resid"].float()
    if has("resid_f32"): v = v + t["resid_f32"]
    v = v * scale
    if act == 1: v = torch.nn.functional.silu(v)
    if act == 2: v = torch.nn.functional.gelu(v, approximate="tanh")
    return c.float(), v, part, bm.value


CASES = [   # (variant, M, N, K, log2 rows per sample, fused terms, activation, fp32 output)
    (V_DMA256P, 1000, 392, 192, 8, ("bias_n",), 0, False),                       # EPI 1, ragged M and N
    (V_DMA256P, 1024, 512, 128, 8, ("bias_n", "rowvec", "gn"), 0, False),        # EPI 2
    (V_DMA128P, 900, 264, 64, 7, ("bias_n",), 1, False),                         # EPI 3 (SiLU)
    (V_DMA256P, 777, 1160, 256, 8, ("bias_n",), 2, False),                       # EPI 4 (tanh-GELU)
    (V_RING256W4, 1000, 136, 128, 8, ("bias_n", "resid"), 0, False),             # EPI 5
    (V_DMA512, 2048, 128, 1152, 9, ("bias_n", "rowvec", "resid", "gn"), 0, False),   # EPI 6 on the 512x128 tile
    (V_DMA256P, 1000, 392, 192, 30, ("bias_n", "gate", "resid_f32"), 0, True),   # EPI 7 (direct fp32 residual stream)
    (V_RING64, 333, 1536, 128, 30, ("bias_n", "gate", "resid_f32"), 0, True),    # EPI 7, context-stream shape
    (V_DMA128P, 520, 328, 128, 8, ("bias_m",), 0, False),                        # EPI 8 (row bias)
    (V_DMA128P, 512, 256, 128, 6, ("bias_n", "rowvec"), 0, False),               # two samples per tile -> general epilogue
    (V_DMA256P, 2200, 2568, 128, 30, ("bias_n",), 0, False),                     # 9 x 11 tiles: grouped rasterisation, partial last group
    (V_DMA128P, 2100, 1160, 64, 30, ("bias_n", "gate", "resid_f32"), 0, True),   # 17 x 10 tiles of 128: grouped rasterisation
    (V_W128, 1000, 392, 192, 8, ("bias_n",), 0, False),                          # EPI 1, ragged M and N, three K-tiles (one steady-state iteration)
    (V_W128, 1024, 512, 128, 8, ("bias_n", "rowvec", "gn"), 0, False),           # GroupNorm partials (general epilogue on this tile), two K-tiles (no steady-state iteration)
    (V_W128, 776, 1160, 256, 8, ("bias_n",), 2, False),                          # EPI 4, second half tile ragged
    (V_W128, 1000, 136, 1536, 8, ("bias_n", "resid"), 0, False),                 # EPI 5, 24 K-tiles, second half tile of eight columns
    (V_W128, 2048, 128, 1152, 9, ("bias_n", "rowvec", "resid", "gn"), 0, False), # residual + partials (general epilogue), no second half tile
    (V_W128, 1000, 392, 192, 30, ("bias_n", "gate", "resid_f32"), 0, True),      # EPI 7
    (V_W128, 520, 328, 128, 8, ("bias_m",), 0, False),                           # EPI 8
    (V_W128, 2200, 2568, 320, 30, ("bias_n",), 0, False),                        # 9 x 11 tiles, five K-tiles
    (V_W128, 4096, 1536, 6144, 30, ("bias_n",), 0, False),                       # 96 K-tiles, 96 tiles
]


@pytest.mark.parametrize("variant,M,N,K,lrs,terms,act,c_f32", CASES)
def test_epilogue_matches_fp32_reference(variant, M, N, K, lrs, terms, act, c_f32):
    out, ref, part, bm = _run(variant, M, N, K, lrs, terms, act, c_f32)
    assert torch.isfinite(out).all()
    tol = (2e-6 if c_f32 else 2 ** -8) * ref.abs().max().item() + 1e-5
    assert (out - ref).abs().max().item() <= tol
    # the general epilogue computes the same thing (not bit-identical: fma contraction differs)
    out0, _, part0, bm0 = _run(variant, M, N, K, lrs, terms, act, c_f32, fp32_slab=True)
    assert (out0 - ref).abs().max().item() <= tol
    if part is not None:
        assert bm == bm0 and M % bm == 0
        want = ref.reshape(M // bm, bm, N // 4, 4)
        s, q = want.sum(dim=(1, 3)), (want * want).sum(dim=(1, 3))
        got = part[: (M // bm) * (N // 4) * 2].reshape(M // bm, N // 4, 2)
        got0 = part0[: (M // bm) * (N // 4) * 2].reshape(M // bm, N // 4, 2)
        for g_ in (got, got0):
            assert (g_[..., 0] - s).abs().max().item() <= 2e-3 * q.sqrt().max().item() + 1e-3
            assert (g_[..., 1] - q).abs().max().item() <= 2e-3 * q.max().item()


def test_w128_tile_equals_the_eight_wave_tile():
    """k_gemm_w128 (one wave per SIMD, 128 x 128 wave tiles, AGPR accumulators) adds the same K steps in the same order as k_gemm_dma<2, 4, 8, 4, 6>: the two
    tiles' bf16 outputs are the same bytes, fused terms included; and the dispatcher takes it for a plain long-K GEMM unless natinf_set_gemm_w128(0)"""
    from naturaldiffusion_amd._lib import lib, check
    for terms, act, c_f32 in ((("bias_n",), 0, False), (("bias_n",), 2, False), (("bias_n", "gate", "resid_f32"), 0, True)):
        a = _run(V_W128, 2048, 1536, 1536, 30, terms, act, c_f32)
        b = _run(V_DMA256P, 2048, 1536, 1536, 30, terms, act, c_f32)
        assert torch.equal(a[0], b[0])
        auto = _run(0, 65536, 1536, 1536, 30, terms, act, c_f32, seed=1)            # 1,536 tiles: the automatic choice ...
        check(lib.natinf_set_gemm_w128(0), "set")
        try:
            old = _run(0, 65536, 1536, 1536, 30, terms, act, c_f32, seed=1)         # ... and the pre-round-4 one
        finally:
            check(lib.natinf_set_gemm_w128(1), "set")
        assert torch.equal(auto[0], old[0])


def test_w128_tile_with_a_half_tile_of_columns():
    """N % 256 == 128 on the four-wave tile (round 5: DiT-XL/2's q | k | v projection, (4096, 3456, 1152) = 13.5 column tiles; the kernel skips the half that lies
    beyond N): forced and as the automatic choice, against the fp32 reference and -- the same K steps in the same order -- the 128 x 128 tiles' bytes"""
    from naturaldiffusion_amd._lib import lib, check
    for (M, N, K) in ((4096, 3456, 1152), (1024, 384, 1024)):
        for terms, act in ((("bias_n",), 0), (("bias_n",), 2)):
            forced = _run(V_W128, M, N, K, 30, terms, act, False)
            auto = _run(0, M, N, K, 30, terms, act, False)
            check(lib.natinf_set_gemm_w128(0), "set")
            try:
                old = _run(0, M, N, K, 30, terms, act, False)
            finally:
                check(lib.natinf_set_gemm_w128(1), "set")
            ref = forced[1]
            assert torch.isfinite(forced[0].float()).all()
            assert (forced[0].float() - ref).abs().max().item() <= 1e-2 * ref.abs().max().item()
            assert torch.equal(forced[0], old[0]) and torch.equal(auto[0], old[0])


def test_w128_splitk_with_the_gated_fp32_residual_epilogue():
    """an under-filled long-K GEMM with the transformer engines' residual epilogue (DiT-XL/2's fc2 at 16 samples: 80 tiles of 256 x 256, 72 K-tiles): K slices on
    k_gemm_w128<9> + k_splitk_reduce_f32 against the fp32 reference and against the unsplit launch; deterministic; the in-place form (residual == output) is what the
    engines run"""
    for (M, N, K, lrs) in ((4096, 1152, 4608, 8), (1000, 392, 3072, 30), (2048, 1152, 4608, 8)):
        terms = ("bias_n", "gate", "resid_f32")
        out, ref, _, bm = _run(0, M, N, K, lrs, terms, 0, True, splitk=4)
        out2, _, _, _ = _run(0, M, N, K, lrs, terms, 0, True, splitk=4)
        out1, _, _, _ = _run(0, M, N, K, lrs, terms, 0, True)
        tol = 2e-6 * ref.abs().max().item() + 1e-5
        assert bm == 256 and torch.isfinite(out).all() and torch.equal(out, out2)
        assert (out - ref).abs().max().item() <= tol and (out1 - ref).abs().max().item() <= tol


def test_packed_epilogue_is_deterministic():
    a = _run(V_DMA256P, 1024, 512, 128, 8, ("bias_n", "rowvec", "gn"))
    b = _run(V_DMA256P, 1024, 512, 128, 8, ("bias_n", "rowvec", "gn"))
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])


SPLITK_CASES = [   # (M, N, K, log2 rows per sample, fused terms): the shapes of the 8x8 / 4x4 levels and ragged relatives
    (2048, 256, 2304, 6, ("bias_n", "rowvec", "gn")),               # Conv_0 at 8x8, 32 samples: 32 tiles -> 4 slices
    (512, 256, 4608, 4, ("bias_n", "resid", "gn")),                 # Conv_1 at 4x4 with the residual
    (1000, 136, 2560, 30, ("bias_n", "rowvec", "resid")),           # ragged M and N (no statistics), N / 8 = 17 does not divide 256 -> not split
    (1000, 128, 2560, 30, ("bias_n", "rowvec", "resid")),           # ragged M
]


@pytest.mark.parametrize("M,N,K,lrs,terms", SPLITK_CASES)
def test_splitk_matches_fp32_reference_and_the_unsplit_launch(M, N, K, lrs, terms):
    out, ref, part, bm = _run(0, M, N, K, lrs, terms, splitk=4)
    out1, _, part1, bm1 = _run(0, M, N, K, lrs, terms)
    tol = 2 ** -8 * ref.abs().max().item() + 1e-5
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max().item() <= tol and (out1 - ref).abs().max().item() <= tol
    assert (out - out1).abs().max().item() <= 2 ** -7 * ref.abs().max().item()       # one bf16 rounding each, fp32 sums in another order
    if N % 8 == 0 and 256 % (N // 8) == 0:
        assert bm == 16
    if part is not None:
        assert M % bm == 0
        want = ref.reshape(M // bm, bm, N // 4, 4)
        s, q = want.sum(dim=(1, 3)), (want * want).sum(dim=(1, 3))
        got = part[: (M // bm) * (N // 4) * 2].reshape(M // bm, N // 4, 2)
        assert (got[..., 0] - s).abs().max().item() <= 2e-3 * q.sqrt().max().item() + 1e-3
        assert (got[..., 1] - q).abs().max().item() <= 2e-3 * q.max().item()
    again = _run(0, M, N, K, lrs, terms, splitk=4)
    assert torch.equal(again[0], out)                                  # slices are summed in a fixed order


def test_gemm_profile_tags_every_launch_where_it_runs():
    """natinf_gemm_profile / natinf_gemm_profile_read (round 6; include/natinf_ncsnpp.h): while enabled, every matmul-shaped launch is bracketed by a HIP event pair on its
    stream and tagged with its launch description; the read sums per tag, in first-seen order, and consumes the records.  The results must not depend on the switch."""
    from naturaldiffusion_amd._lib import lib, check
    ref = _run(V_W128, 512, 256, 256, 30, {"bias_n"}, seed=3)[0].clone()
    buf = C.create_string_buffer(1 << 14)
    try:
        check(lib.natinf_gemm_profile(1), "profile on")
        for _ in range(3):
            got = _run(V_W128, 512, 256, 256, 30, {"bias_n"}, seed=3)[0]
        _run(V_DMA128P, 300, 136, 192, 30, {"bias_n", "resid_f32"}, c_f32=True)
        torch.cuda.synchronize()
    finally:
        check(lib.natinf_gemm_profile(0), "profile off")
    n = lib.natinf_gemm_profile_read(buf, len(buf))
    assert n > 0, n
    rows = [r.split() for r in buf.value.decode().splitlines()]
    assert [r[:3] for r in rows] == [["512", "256", "256"], ["300", "136", "192"]], rows            # first-seen order, one row per distinct description
    assert rows[0][6].startswith("w128_256x256/e") and int(rows[0][7]) == 3 and float(rows[0][8]) > 0.0
    assert rows[1][6] == "dma128x128p/e7" and int(rows[1][7]) == 1 and float(rows[1][8]) > 0.0      # the direct fp32 residual epilogue, named
    assert lib.natinf_gemm_profile_read(buf, len(buf)) == 0 and buf.value == b""                    # consumed
    assert torch.equal(got, ref)
    _run(V_W128, 512, 256, 256, 30, {"bias_n"}, seed=3)                                              # switched off: nothing is recorded
    assert lib.natinf_gemm_profile_read(buf, len(buf)) == 0
