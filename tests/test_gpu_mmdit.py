"""HIP MMDiT engine (include/natinf_mmdit.h) against oracle/mmdit_oracle.py -- a restatement of the published SD3
architecture; PARITY UNPINNED with respect to the reference's un-vendored ``diffusers`` (see the oracle's header)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 3e-2          # max |engine - oracle| / max |oracle|, bf16 operands vs fp32 oracle


@pytest.fixture(params=[0, 1, 2, 3], ids=["online", "deferred", "deferred+mfma_sums", "deferred+mfma_sums_64key"])
def flash_mode(request):
    """every head_dim-64 attention kernel of the library (natinf_set_flash_mode): the parity cases run on all of them"""
    from naturaldiffusion_amd._lib import lib, check
    rc = lib.natinf_set_flash_mode(request.param)
    if rc == -4 and request.param in (1, 2):                # NATINF_ESTATE
        pytest.skip("modes 1 / 2 (intermediate forms) exist in -DNATINF_DEV builds only")
    check(rc, "natinf_set_flash_mode")
    yield request.param
    check(lib.natinf_set_flash_mode(3), "natinf_set_flash_mode")              # the library's default


def _softmax_ref64(q, k, v, B, T, H):
    hd = lambda t: t.double().reshape(B, T, H, 64).transpose(1, 2)
    w = torch.softmax(hd(q) @ hd(k).transpose(-2, -1) * 0.125, dim=-1)
    return (w @ hd(v)).transpose(1, 2).reshape(B, T, H * 64)


# (285 = 256 + 29 tokens: 99 padded keys -- more than one 64-key tile: the padding reaches into the last TWO tiles of the 64-key kernel; 129: 127 padded keys)
@pytest.mark.parametrize("B,T,H", [(2, 128, 2), (1, 333, 3), (2, 4429, 2), (1, 77, 1), (2, 285, 4), (1, 129, 1), (1, 65, 2), (1, 64, 1)])
def test_flash_attention_matches_fp32_softmax(B, T, H, flash_mode):
    from naturaldiffusion_amd.mmdit import attention_hd64
    g = torch.Generator().manual_seed(T)
    q, k, v = (torch.randn(B, T, H * 64, generator=g).bfloat16() for _ in range(3))
    q = q * 2.0                                                         # sharper softmax: exercises the running-max rescale
    o = attention_hd64(q.cuda(), k.cuda(), v.cuda()).double().cpu()
    ref = _softmax_ref64(q, k, v, B, T, H)
    assert torch.isfinite(o).all()
    assert ((o - ref).abs().max() / ref.abs().max()).item() <= 2e-2


@pytest.mark.parametrize("spike_tile,spike", [(0, 40.0), (3, 40.0), (7, 200.0), (3, -60.0)])
def test_flash_attention_rereferencing_branch(spike_tile, spike, flash_mode):
    """The deferred kernels move their reference only when a score exceeds it by 2^8 -- a rare, data-dependent branch that bounded random data
    never takes after the first tile.  Force it: one key row (in key tile `spike_tile`) is aligned with every query so that its scores
    jump ~40 / ~200 exponent units above everything before it (both groups of a wave re-reference at that tile; with 200 the old
    accumulators are scaled by 2^-200: flushed); a NEGATIVE alignment (scores far below the reference) must NOT trigger anything and must
    not underflow the rest.  Also: queries whose first tile holds only very negative scores (reference below zero).  fp64 reference."""
    from naturaldiffusion_amd.mmdit import attention_hd64
    B, T, H = 1, 1100, 2
    g = torch.Generator().manual_seed(17 + spike_tile)
    q, k, v = (torch.randn(B, T, H * 64, generator=g) for _ in range(3))
    u = torch.randn(64, generator=g); u = u / u.norm()
    qh = q.view(B, T, H, 64); kh = k.view(B, T, H, 64)
    qh += 6.0 * u                                                        # every query has a component of ~6 along u ...
    kh[:, 128 * spike_tile + 37] = (spike * 8.0 / 6.0 / 0.125 / 1.4427 / 8.0) * u      # ... and this key scores ~spike log2-units against it
    kh[:, :128] = 0.1 * kh[:, :128] - 8.0 * u                            # the whole first key tile scores BELOW zero for every query: the first reference is negative
    q, k, v = q.bfloat16(), k.bfloat16(), v.bfloat16()
    o = attention_hd64(q.cuda(), k.cuda(), v.cuda()).double().cpu()
    ref = _softmax_ref64(q, k, v, B, T, H)
    assert torch.isfinite(o).all()
    assert ((o - ref).abs().max() / ref.abs().max()).item() <= 2e-2


def _small():
    from oracle import mmdit_oracle as M
    cfg = dict(layers=3, heads=2, joint_dim=64, pooled_dim=32)
    P = M.make_params(seed=4, pos_max=24, pos_base=8, **cfg)
    return M, cfg, P


def test_small_config_matches_oracle_and_batch_independent():
    from naturaldiffusion_amd.mmdit import MMDiTEngine, flatten_state_dict
    M, cfg, P = _small()
    grid, tc = 8, 13
    eng = MMDiTEngine(flatten_state_dict(P, grid, **cfg), max_batch=4, grid=grid, ctx_tokens=tc, **cfg)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 16, 16, 16, generator=g)
    t = torch.tensor([900.0, 10.0, 455.5])
    e = torch.randn(3, tc, 64, generator=g)
    p = torch.randn(3, 32, generator=g)
    taps = {}
    ref = M.forward(P, x, t, e, p, taps=taps)
    out = eng.forward(x.cuda(), t.cuda(), e.cuda(), p.cuda()).cpu()
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    assert err <= TOL, err
    # one sequence alone gives the same answer as inside the batch
    solo = eng.forward(x[1:2].cuda(), t[1:2].cuda(), e[1:2].cuda(), p[1:2].cuda()).cpu()
    assert ((solo[0] - out[1]).abs().max() / ref.abs().max()).item() <= 1e-2
    # call shape of pipe.transformer (src/SD3NaturalInference.py:210-213), fp16 in / fp16 out
    o16 = eng(hidden_states=x.half().cuda(), timestep=t.cuda(), encoder_hidden_states=e.half().cuda(), pooled_projections=p.half().cuda(),
              return_dict=False)[0]
    assert o16.dtype == torch.float16 and o16.shape == x.shape


def test_wider_config_matches_oracle():
    """D = 384 (6 heads), text tokens not a multiple of 8, joint sequence 64+77=141 -> padded to 256."""
    from oracle import mmdit_oracle as M
    from naturaldiffusion_amd.mmdit import MMDiTEngine, flatten_state_dict
    cfg = dict(layers=2, heads=6, joint_dim=256, pooled_dim=128)
    P = M.make_params(seed=8, pos_max=16, pos_base=8, **cfg)
    eng = MMDiTEngine(flatten_state_dict(P, 8, **cfg), max_batch=2, grid=8, ctx_tokens=77, **cfg)
    g = torch.Generator().manual_seed(1)
    x, t = torch.randn(2, 16, 16, 16, generator=g), torch.tensor([1000.0, 1.0])
    e, p = torch.randn(2, 77, 256, generator=g), torch.randn(2, 128, generator=g)
    ref = M.forward(P, x, t, e, p)
    out = eng.forward(x.cuda(), t.cuda(), e.cuda(), p.cuda()).cpu()
    assert ((out - ref).abs().max() / ref.abs().max()).item() <= TOL


def test_argument_errors():
    from naturaldiffusion_amd.mmdit import MMDiTEngine, flatten_state_dict
    M, cfg, P = _small()
    flat = flatten_state_dict(P, 8, **cfg)
    with pytest.raises(ValueError):
        MMDiTEngine(flat[:-1], max_batch=1, grid=8, ctx_tokens=13, **cfg)
    eng = MMDiTEngine(flat, max_batch=1, grid=8, ctx_tokens=13, **cfg)
    z = torch.zeros(2, 16, 16, 16).cuda()
    with pytest.raises(ValueError):
        eng.forward(z, torch.zeros(2), torch.zeros(2, 13, 64), torch.zeros(2, 32))          # batch > max_batch
    with pytest.raises(ValueError):
        eng.forward(z[:1], torch.zeros(1), torch.zeros(1, 12, 64), torch.zeros(1, 32))      # wrong text length


def test_sd3_natural_inference_with_the_engine_as_pipe_transformer(repo_root):
    """src/SD3NaturalInference.py:172-245 with ``pipe.transformer`` = the HIP engine (text + null prompts batched into
    one forward per step), against the oracle's restatement of the loop driven by the MMDiT oracle.  fp16 chain with a
    bf16-operand denoiser vs an fp32 one: tolerance on the final latents, not bit equality."""
    from oracle import ni_oracle as O
    from naturaldiffusion_amd import SD3NaturalInference as S
    from naturaldiffusion_amd.mmdit import MMDiTEngine, flatten_state_dict
    M, cfg, P = _small()
    n, grid, tc = 2, 8, 13
    # a denoiser with O(1) velocities of the right sign keeps the 28-step fp16 chain well conditioned: scale the output layer
    P = dict(P); P["proj_out.weight"] = P["proj_out.weight"] * 0.2
    eng = MMDiTEngine(flatten_state_dict(P, grid, **cfg), max_batch=2 * n, grid=grid, ctx_tokens=tc, **cfg)
    g = torch.Generator().manual_seed(3)
    pe, ne = torch.randn(n, tc, 64, generator=g).half(), torch.randn(n, tc, 64, generator=g).half()
    ppe, npe = torch.randn(n, 32, generator=g).half(), torch.randn(n, 32, generator=g).half()
    noises = torch.randn(n, 16, 16, 16, generator=g).half()

    class Sched:
        def set_timesteps(self, k, device=None):
            self.timesteps, self.sigmas = O.sd3_sigma_schedule(k)

    class Pipe:
        scheduler = Sched()
        transformer = eng

        def encode_prompt(self, prompt, **k):
            return (pe.cuda(), ne.cuda(), ppe.cuda(), npe.cuda())
    finals = S.sd_natural_inference_tx(pipe=Pipe(), device="cuda:0", noises=noises.cuda(), n=n, decode=False,
                                       weight_names=("sd3_step_28_weight.csv",))
    W = O.load_sd3_csv(repo_root / "weights/sd3_step_28_weight.csv")
    ts, sig = O.sd3_sigma_schedule(28)

    def vel(x, t, cond):
        tt = torch.as_tensor(t, dtype=torch.float32).expand(n)
        return M.forward(P, x.float(), tt, (pe if cond else ne).float(), (ppe if cond else npe).float()).half()
    ref = O.sd3_ni(vel, noises, W, sig, ts)
    out = finals[0].cpu().float()
    assert torch.isfinite(out).all()
    assert ((out - ref.float()).abs().max() / ref.float().abs().max()).item() <= 5e-2


def test_full_sequence_length_narrow_width_matches_oracle():
    """The SD3 token counts (64x64 image tokens + 333 text tokens = 4,429 -> padded to 4,480) at a narrow width, so that
    the CPU oracle finishes in seconds: exercises the joint-buffer offsets, the key-tail mask and the 35-tile key walk."""
    from oracle import mmdit_oracle as M
    from naturaldiffusion_amd.mmdit import MMDiTEngine, flatten_state_dict
    cfg = dict(layers=2, heads=2, joint_dim=64, pooled_dim=32)
    P = M.make_params(seed=6, pos_max=192, pos_base=64, **cfg)
    eng = MMDiTEngine(flatten_state_dict(P, 64, **cfg), max_batch=2, grid=64, ctx_tokens=333, **cfg)
    g = torch.Generator().manual_seed(2)
    x, t = torch.randn(2, 16, 128, 128, generator=g), torch.tensor([700.0, 30.0])
    e, p = torch.randn(2, 333, 64, generator=g), torch.randn(2, 32, generator=g)
    ref = M.forward(P, x, t, e, p)
    out = eng.forward(x.cuda(), t.cuda(), e.cuda(), p.cuda()).cpu()
    assert torch.isfinite(out).all()
    assert ((out - ref).abs().max() / ref.abs().max()).item() <= TOL


@pytest.mark.parametrize("fp8", [False, True])
def test_forward_is_deterministic_and_ignores_workspace_contents(fp8):
    """Two forwards over a NaN-poisoned and a zeroed workspace give bit-identical, finite results (guards the LDS-DMA
    completion wait in the attention kernel and every padded / never-written region of the joint buffers)."""
    from oracle import mmdit_oracle as M
    from naturaldiffusion_amd.mmdit import MMDiTEngine, flatten_state_dict
    cfg = dict(layers=4, heads=4, joint_dim=64, pooled_dim=32)
    P = M.make_params(seed=12, pos_max=192, pos_base=64, **cfg)
    eng = MMDiTEngine(flatten_state_dict(P, 64, **cfg), max_batch=3, grid=64, ctx_tokens=333, fp8=fp8, **cfg)
    g = torch.Generator().manual_seed(4)
    x, t = torch.randn(3, 16, 128, 128, generator=g).cuda(), torch.tensor([999.0, 500.0, 1.0]).cuda()
    e, p = torch.randn(3, 333, 64, generator=g).cuda(), torch.randn(3, 32, generator=g).cuda()
    eng._ws.view(torch.int16).fill_(0x7FC0)                       # bf16 NaN in every 2 bytes (fp32 NaN as well)
    a = eng.forward(x, t, e, p)
    eng._ws.zero_()
    b = eng.forward(x, t, e, p)
    assert torch.isfinite(a).all() and torch.equal(a, b)
    # FINITE garbage (what an allocator block of an earlier engine holds): NaN cannot stand in for it -- a NaN compares false, so a padded query row of NaNs never
    # tripped the attention kernel's wave-wide re-referencing test, while finite left-overs did and moved the real rows' last bits (round 4's red test: the padded
    # q | k rows of the joint buffer were never written; tools/diag_fp8_ws.py bisected the workspace to them)
    junk = torch.randint(0, 0x60, (eng._ws.numel(),), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    eng._ws.copy_(junk)
    c = eng.forward(x, t, e, p)
    assert torch.equal(c, b)


@pytest.fixture(params=[0, 1, 2], ids=["eight_wave_tile", "w128", "w128_except_fc1"])
def fp8_tile(request):
    """natinf_set_gemm_w128: 0 = k_gemm_fp8 (eight waves, two per SIMD) for every launch, 1 (the default) = k_gemm_w128_fp8 (four waves, 128 x 128 wave tiles) where the
    K-tile count is even, the e4m3 + E8M0 output epilogue included (round 5); 2 = all but that one (the round-4 rule)"""
    from naturaldiffusion_amd._lib import lib, check
    check(lib.natinf_set_gemm_w128(request.param), "set")
    yield request.param
    check(lib.natinf_set_gemm_w128(1), "set")


@pytest.mark.parametrize("fp8", [False, True])
def test_text_stream_on_its_own_hip_stream_gives_the_same_bytes(fp8):
    """natinf_set_mmdit_text_stream: the text stream's launches on the engine's second HIP stream (fork / join events around every joint attention) against
    everything on the caller's stream -- same launches, same arguments, so the same bytes; repeated forwards (the events are re-recorded) and a forward on a
    non-default caller stream included."""
    from naturaldiffusion_amd._lib import lib, check
    from naturaldiffusion_amd.mmdit import MMDiTEngine
    from naturaldiffusion_amd.synth import synthetic_mmdit_flat
    cfg = dict(layers=3, heads=4, joint_dim=64, pooled_dim=32, in_ch=16)
    eng = MMDiTEngine(synthetic_mmdit_flat(8, seed=3, **cfg), max_batch=3, grid=8, ctx_tokens=37, fp8=fp8, **cfg)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(3, 16, 16, 16, device="cuda", generator=g); t = torch.rand(3, device="cuda", generator=g) * 1000
    e = torch.randn(3, 37, 64, device="cuda", generator=g); p = torch.randn(3, 32, device="cuda", generator=g)
    try:
        check(lib.natinf_set_mmdit_text_stream(0), "set")
        ref = eng.forward(x, t, e, p)
        check(lib.natinf_set_mmdit_text_stream(1), "set")
        for _ in range(3):
            assert torch.equal(eng.forward(x, t, e, p), ref)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                out = eng.forward(x, t, e, p)
        side.synchronize()
        assert torch.equal(out, ref) and torch.isfinite(ref).all()
    finally:
        check(lib.natinf_set_mmdit_text_stream(1), "set")


def test_fp8_gemm_matches_the_dequantised_product(fp8_tile):
    """k_gemm_fp8 / k_gemm_w128_fp8 on their own: same quantised operands, fp32 reference -> only the bf16 output rounding remains."""
    from naturaldiffusion_amd._lib import lib, check, ptr, stream_ptr
    g = torch.Generator().manual_seed(3)
    # K = 1536 / 6144: the SD3-medium projections (q|k|v, fc1 / fc2): 12 and 48 K-tiles through the two-stage DMA pipeline
    for (M, N, K) in ((512, 256, 256), (1000, 520, 384), (300, 264, 128), (768, 1536, 1536), (520, 1536, 6144), (1000, 392, 512), (264, 1288, 768)):      # (the last two: ragged tiles on the w128 kernel)
        a = (torch.randn(M, K, generator=g) * (torch.rand(M, 1, generator=g) * 3 + 0.1)).cuda()
        b = (torch.randn(N, K, generator=g) * 0.05).cuda()
        bias = torch.randn(N, generator=g).cuda()
        qa, sa = torch.empty(M, K, dtype=torch.uint8, device="cuda"), torch.empty(M, device="cuda")
        qb, sb = torch.empty(N, K, dtype=torch.uint8, device="cuda"), torch.empty(N, device="cuda")
        check(lib.natinf_debug_quant_fp8_rows(ptr(a), ptr(qa), ptr(sa), M, K, stream_ptr()), "quant")
        check(lib.natinf_debug_quant_fp8_rows(ptr(b), ptr(qb), ptr(sb), N, K, stream_ptr()), "quant")
        c = torch.empty(M, N, device="cuda")
        check(lib.natinf_debug_gemm_fp8(M, N, K, ptr(qa), ptr(sa), None, ptr(qb), ptr(sb), ptr(bias), ptr(c), None, 1, 1, stream_ptr()), "gemm_fp8")
        da = qa.view(torch.float8_e4m3fn).float() * sa[:, None]
        db = qb.view(torch.float8_e4m3fn).float() * sb[:, None]
        ref = (da.double() @ db.double().t() + bias.double()).float()
        assert ((c - ref).abs().max() / ref.abs().max()).item() <= 1e-4          # fp32 accumulation order only
        # the quantiser itself: per-row scale = max|row| / 448, values within e4m3 rounding (2^-3 relative) of the input
        assert torch.allclose(sa, a.abs().amax(dim=1) / 448.0, rtol=1e-6)
        assert ((da - a).abs() <= 0.0625 * a.abs() + sa[:, None] * 2.0 ** -9 + 1e-12).all()


def _mx_quant(x):
    """reference MX quantiser: per row and 32 columns, scale 2^e = the smallest power of two with amax * 2^-e <= 448."""
    M, K = x.shape
    b = x.reshape(M, K // 32, 32)
    amax = b.abs().amax(dim=2)
    e = torch.ceil(torch.log2(amax.clamp_min(1e-38) / 448.0)).clamp(-127, 127)
    e = torch.where(amax > 0, e, torch.zeros_like(e))
    q = (b * torch.exp2(-e)[..., None]).clamp(-448, 448).to(torch.float8_e4m3fn)
    return q.reshape(M, K).view(torch.uint8).contiguous(), (e + 127).to(torch.uint8).contiguous()


def _mx_dequant(q, sc):
    M, K = q.shape
    return (q.view(torch.float8_e4m3fn).float().reshape(M, K // 32, 32) * torch.exp2(sc.float() - 127)[..., None]).reshape(M, K)


def _ktile_major(sc):
    """[M][K/32] scale bytes -> the kernel's K-tile-major layout [(K/128)][M][4] (flattened), padded by 1 KiB"""
    M, nb = sc.shape
    t = sc.reshape(M, nb // 4, 4).permute(1, 0, 2).contiguous().reshape(-1)
    return torch.cat([t, torch.full((1024,), 127, dtype=torch.uint8, device=sc.device)])


def test_fp8_gemm_with_mx_block_scales_in_and_out(fp8_tile):
    """A operand with E8M0 block scales per 32 K-elements (fed to the MFMA lane by lane), and the fp8 + block-scale OUTPUT
    mode of the epilogue (what fc1 hands to fc2 in the fp8 engine)."""
    from naturaldiffusion_amd._lib import lib, check, ptr, stream_ptr
    g = torch.Generator().manual_seed(5)
    for (M, N, K) in ((512, 256, 256), (700, 640, 384), (512, 1536, 6144)):
        a = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, K // 32, 1, generator=g)).expand(M, K // 32, 32).reshape(M, K)).cuda()
        b = (torch.randn(N, K, generator=g) * 0.05).cuda()
        qa, ma = _mx_quant(a)
        qb, sb = torch.empty(N, K, dtype=torch.uint8, device="cuda"), torch.empty(N, device="cuda")
        check(lib.natinf_debug_quant_fp8_rows(ptr(b), ptr(qb), ptr(sb), N, K, stream_ptr()), "quant")
        ref = (_mx_dequant(qa, ma).double() @ (qb.view(torch.float8_e4m3fn).float() * sb[:, None]).double().t()).float()
        c = torch.empty(M, N, device="cuda")
        mat = _ktile_major(ma)
        check(lib.natinf_debug_gemm_fp8(M, N, K, ptr(qa), None, ptr(mat), ptr(qb), ptr(sb), None, ptr(c), None, 1, 1, stream_ptr()), "gemm_fp8")
        assert ((c - ref).abs().max() / ref.abs().max()).item() <= 1e-4
        assert ((_mx_dequant(qa, ma) - a).abs() <= 0.0625 * a.abs() + 1e-6 * a.abs().max()).all()      # the reference quantiser itself
        # output mode: e4m3 bytes + one scale per row and 32 columns, against the fp32 output of the same launch
        c8, cmt = torch.empty(M, N, dtype=torch.uint8, device="cuda"), torch.empty((N // 128) * M * 4, dtype=torch.uint8, device="cuda")
        check(lib.natinf_debug_gemm_fp8(M, N, K, ptr(qa), None, ptr(mat), ptr(qb), ptr(sb), None, ptr(c8), ptr(cmt), 3, 1, stream_ptr()), "gemm_fp8")
        cm = cmt.reshape(N // 128, M, 4).permute(1, 0, 2).reshape(M, N // 32).contiguous()
        want_q, want_m = _mx_quant(c)
        assert torch.equal(cm, want_m)
        dec = _mx_dequant(c8, cm)
        blockmax = c.reshape(M, N // 32, 32).abs().amax(dim=2, keepdim=True).expand(M, N // 32, 32).reshape(M, N)
        assert ((dec - c).abs() <= 0.0625 * c.abs() + blockmax * 2.0 ** -9 + 1e-12).all()
        # the fc1 epilogue itself (c_mode 3 | tanh-GELU << 8: k_gemm_fp8<., 2> / k_gemm_w128_fp8<., 2>, the staged eight-value GELU): against
        # GELU of the fp32 output of the same product.  The kernel's exp2 / rcp differ from torch's tanh in the last bits, so a block whose maximum sits
        # on a power-of-two boundary may take the neighbouring scale: at most one step, in a handful of blocks.
        gl = torch.nn.functional.gelu(c, approximate="tanh")
        check(lib.natinf_debug_gemm_fp8(M, N, K, ptr(qa), None, ptr(mat), ptr(qb), ptr(sb), None, ptr(c8), ptr(cmt), 3 | (2 << 8), 1, stream_ptr()), "gemm_fp8")
        cm = cmt.reshape(N // 128, M, 4).permute(1, 0, 2).reshape(M, N // 32).contiguous()
        _, want_m = _mx_quant(gl)
        diff = (cm.int() - want_m.int()).abs()
        assert diff.max().item() <= 1 and (diff != 0).float().mean().item() <= 1e-3
        dec = _mx_dequant(c8, cm)
        blockmax = gl.reshape(M, N // 32, 32).abs().amax(dim=2, keepdim=True).expand(M, N // 32, 32).reshape(M, N)
        assert ((dec - gl).abs() <= 0.0625 * gl.abs() + blockmax * 2.0 ** -8 + 1e-6 * gl.abs().max()).all()


def test_fp8_engine_close_to_bf16_engine_and_oracle():
    """NATINF_MMDIT_FP8 (config 5): image-stream q|k, v, fc1 on e4m3 operands.  Tolerance is the fp8 one (3 significant
    bits per operand element): 8e-2 of the output's max magnitude against the fp32 oracle."""
    from oracle import mmdit_oracle as M
    from naturaldiffusion_amd.mmdit import MMDiTEngine, flatten_state_dict
    cfg = dict(layers=3, heads=2, joint_dim=64, pooled_dim=32)
    P = M.make_params(seed=4, pos_max=24, pos_base=8, **cfg)
    flat = flatten_state_dict(P, 8, **cfg)
    g = torch.Generator().manual_seed(0)
    x, t = torch.randn(3, 16, 16, 16, generator=g), torch.tensor([900.0, 10.0, 455.5])
    e, p = torch.randn(3, 13, 64, generator=g), torch.randn(3, 32, generator=g)
    ref = M.forward(P, x, t, e, p)
    o8 = MMDiTEngine(flat, max_batch=3, grid=8, ctx_tokens=13, fp8=True, **cfg).forward(x.cuda(), t.cuda(), e.cuda(), p.cuda()).cpu()
    o16 = MMDiTEngine(flat, max_batch=3, grid=8, ctx_tokens=13, **cfg).forward(x.cuda(), t.cuda(), e.cuda(), p.cuda()).cpu()
    assert torch.isfinite(o8).all()
    e8 = ((o8 - ref).abs().max() / ref.abs().max()).item()
    e16 = ((o16 - ref).abs().max() / ref.abs().max()).item()
    assert e16 <= TOL and e8 <= 8e-2, (e16, e8)
    with pytest.raises(ValueError):
        MMDiTEngine(flat, max_batch=1, grid=8, ctx_tokens=13, fp8=True, layers=3, heads=3, joint_dim=64, pooled_dim=32)


import functools


@functools.lru_cache(maxsize=1)
def _sd3_width_case():
    from oracle import mmdit_oracle as M
    cfg = dict(layers=2, heads=24, joint_dim=4096, pooled_dim=2048)
    P = M.make_params(seed=21, pos_max=64, pos_base=64, **cfg)
    g = torch.Generator().manual_seed(9)
    x, t = torch.randn(1, 16, 128, 128, generator=g), torch.tensor([640.0])
    e, p = torch.randn(1, 333, 4096, generator=g), torch.randn(1, 2048, generator=g)
    return cfg, P, (x, t, e, p), M.forward(P, x, t, e, p)


@pytest.mark.parametrize("fp8", [False, True])
def test_sd3_medium_width_joint_block_matches_oracle(fp8):
    """BASELINE configs 4 / 5 at the width they are benchmarked on: D = 1536, 24 heads x 64, 64x64 image tokens + 333
    text tokens (4,429 -> padded 4,480 keys), joint_dim 4096, pooled 2048 -- two blocks (one full JointTransformerBlock with
    both MLPs, one context_pre_only), one sequence, so that the CPU oracle (~0.8 TFLOP, 1.9 GB of scores) finishes in well
    under a minute.  Exercises what the narrow configurations cannot: 12- and 48-K-tile GEMM loops (bf16 and fp8), the
    residual-stream / packed / fp8+MX epilogues at N = 1536 / 6144, 24-head flash attention over the joint buffer."""
    from naturaldiffusion_amd.mmdit import MMDiTEngine, flatten_state_dict
    cfg, P, (x, t, e, p), ref = _sd3_width_case()
    eng = MMDiTEngine(flatten_state_dict(P, 64, **cfg), max_batch=1, grid=64, ctx_tokens=333, fp8=fp8, **cfg)
    out = eng.forward(x.cuda(), t.cuda(), e.cuda(), p.cuda()).cpu()
    assert torch.isfinite(out).all()
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    print(f"SD3-medium-width block, fp8={fp8}: max rel err {err:.3e}")
    assert err <= (8e-2 if fp8 else TOL), err
    # the benchmarked batch: the same sequence eight times (every GEMM of the forward batched over eight sequences, the image stream's on the four-wave tiles)
    # must give the one-sequence result eight times
    eng8 = MMDiTEngine(flatten_state_dict(P, 64, **cfg), max_batch=8, grid=64, ctx_tokens=333, fp8=fp8, **cfg)
    rep = lambda v: v.cuda().repeat(8, *([1] * (v.dim() - 1)))
    # (over a workspace of finite left-overs: what this engine gets from the allocator in the middle of a test session)
    eng8._ws.copy_(torch.randint(0, 0x60, (eng8._ws.numel(),), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2)))
    out8 = eng8.forward(rep(x), rep(t), rep(e), rep(p)).cpu()
    # the benchmarked plan against the oracle DIRECTLY, at the same bound as the one-sequence plan
    err8 = ((out8 - ref).abs().amax(dim=(1, 2, 3)) / ref.abs().max()).tolist()
    print("eight sequences vs the oracle:", ["%.3e" % d for d in err8])
    assert max(err8) <= (8e-2 if fp8 else TOL), err8
    # the eight sequences are the same sequence: the same bytes eight times (one launch, one plan, one arithmetic per row)
    assert all(torch.equal(out8[i], out8[0]) for i in range(1, 8))
    # eight against one: the two batches may take different plans (one sequence under-fills the chip: split-K on the long-K GEMMs, the round model's tile choice), i.e.
    # a different order of the fp32 sums over K; measured on this case: identical bytes under every plan knob (tools/diag_fp8_batch.py, profiles/r05/diag_fp8_batch.json).
    # Round 4's 3.6e-3 here was not the plan: the padded query rows of the joint buffer were never written, and finite garbage in them re-referenced the attention
    # kernel's waves that also hold real rows (mmdit_engine.inc: the rows are zeroed per forward now).  Bound: a tenth of the fp8 (bf16) error against the oracle.
    diff = [((out8[i] - out[0]).abs().max() / ref.abs().max()).item() for i in range(8)]
    print("eight sequences vs one:", ["%.2e" % d for d in diff])
    assert max(diff) <= (2e-3 if fp8 else 5e-4), diff


@pytest.mark.parametrize("fp8", [False, True])
def test_half_image_stream_against_the_fp32_stream(fp8):
    """natinf_set_mmdit_stream16 (round 6; the library's default): the image tokens' residual stream in IEEE half -- the reference's SD3 pipeline is fp16 end to end
    (src/SD3NaturalInference.py:175-176) -- against the fp32 stream, at SD3-medium width (the direct residual epilogue on the four-wave tiles at N = 1536, both
    LayerNorm-modulate kernels, the patch embedding): BOTH inside the oracle bound of the test above, and close to each other (every residual update is computed in
    fp32 from the half row and rounded once: 2^-11 per update)."""
    from naturaldiffusion_amd.mmdit import MMDiTEngine, flatten_state_dict
    cfg, P, (x, t, e, p), ref = _sd3_width_case()
    flat = flatten_state_dict(P, 64, **cfg)
    outs, ws = {}, {}
    for s16 in (False, True):
        eng = MMDiTEngine(flat, max_batch=1, grid=64, ctx_tokens=333, fp8=fp8, stream16=s16, **cfg)
        outs[s16] = eng.forward(x.cuda(), t.cuda(), e.cuda(), p.cuda()).cpu()
        ws[s16] = eng.workspace_bytes
        del eng
    assert ws[True] == ws[False] - 64 * 64 * 1536 * 2                     # the image stream is the only buffer that shrinks: 4,096 tokens x 1,536 x 2 bytes per sequence
    for s16 in (False, True):
        assert torch.isfinite(outs[s16]).all()
        err = ((outs[s16] - ref).abs().max() / ref.abs().max()).item()
        print(f"fp8={fp8} stream16={s16}: max rel err against the oracle {err:.3e}")
        assert err <= (8e-2 if fp8 else TOL), (s16, err)
    d = ((outs[True] - outs[False]).abs().max() / ref.abs().max()).item()
    print(f"fp8={fp8}: half stream against fp32 stream {d:.3e}")
    assert d <= (3e-2 if fp8 else 3e-3), d                                 # (fp8: a last-bit change of the stream flips e4m3 roundings downstream -- the scale of fp8's own error)


@pytest.mark.parametrize("csv", ["sd3_step_28_weight.csv", "sd3_step_28_weight_sharp.csv"])
def test_fp8_accuracy_over_a_whole_28_step_run(csv, repo_root):
    """Round-2 review, weak #1: the fp8 path (config 5) had a one-forward bound only.  A whole 28-step SD3-form NI run (CFG 7, the shipped
    coefficient files) at reduced depth -- bf16 engine, fp8 engine, fp32 oracle, identical noise; the numbers bench.py prints in the
    `sd3_fp8.accuracy` object.  Thresholds ~2x the observed values (printed; gpurun_out/sd3_accuracy_<csv>.json)."""
    import json, os, sys
    sys.path.insert(0, str(repo_root))
    from bench import sd3_reduced_depth_accuracy
    rep = sd3_reduced_depth_accuracy(torch.device("cuda:0"), csv)
    os.makedirs(repo_root / "gpurun_out", exist_ok=True)
    (repo_root / "gpurun_out" / f"sd3_accuracy_{csv.split('.')[0]}.json").write_text(json.dumps(rep, indent=1))
    print(json.dumps(rep))
    assert rep["bf16"]["finite"] and rep["fp8"]["finite"]
    # observed (MI355X, round 3): bf16 rel_rms 2.9e-3 / rel_max 3.4-4.5e-3; fp8 rel_rms 1.43-1.46e-2 / rel_max 1.4-1.6e-2
    assert rep["bf16"]["rel_rms"] <= 6e-3 and rep["bf16"]["rel_max"] <= 1e-2, rep
    assert rep["fp8"]["rel_rms"] <= 3e-2 and rep["fp8"]["rel_max"] <= 3.5e-2, rep
    assert rep["fp8"]["rel_rms"] >= rep["bf16"]["rel_rms"] * 0.5                   # fp8 cannot be (much) more accurate than bf16: a sanity check on the comparison
