"""GPU parity of the FID Inception-V3 pool3 engine (include/natinf_inception.h) against oracle/inception_oracle.py (PARITY UNPINNED:
the oracle restates pytorch_fid's published architecture; synthetic weights -- the real ones are a download).  bf16 operands / fp32
accumulate against an fp32 reference: features within 3e-2 of the feature maximum, and -- the quantity that bounds what bf16 can do to
an FID -- the Frechet distance between the statistics of HIP features and oracle features of the same 2,000 synthetic images."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import inception_oracle as O

TOL = 3e-2


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def params():
    return O.make_params(0)


@pytest.fixture(scope="module")
def engine(dev, params):
    from naturaldiffusion_amd.inception import InceptionEngine, flatten_state_dict
    return InceptionEngine(flatten_state_dict(params), max_batch=50, in_hw=(32, 32), device=dev)


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


def test_features_match_the_oracle_for_both_input_forms(engine, params):
    g = torch.Generator().manual_seed(5)
    u8 = torch.randint(0, 256, (6, 32, 32, 3), generator=g, dtype=torch.uint8)
    ref = O.forward(params, u8.permute(0, 3, 1, 2).float() / 255)                  # get_activation's preprocessing (reference :54-56)
    got_u8 = engine(u8).cpu()
    got_f = engine(u8.permute(0, 3, 1, 2).float() / 255).cpu()
    assert torch.isfinite(got_u8).all() and tuple(got_u8.shape) == (6, 2048)
    e = _rel(got_u8, ref)
    print("inception features max-rel error:", e, " mean |d| / mean |ref|:", float((got_u8 - ref).abs().mean() / ref.abs().mean()))
    assert e <= TOL, e
    assert torch.equal(got_u8, got_f)                                              # same arithmetic after the /255
    # batch independence and the chunked walk (max_batch 50): sample 3 alone == sample 3 in the batch; 120 images in three chunks
    assert torch.equal(engine(u8[3:4]).cpu(), got_u8[3:4])
    big = u8.repeat(20, 1, 1, 1)
    assert torch.equal(engine(big).cpu(), got_u8.repeat(20, 1))


def test_other_input_sizes(dev, params):
    """the resize is part of the engine: 64 x 48 inputs (not a CIFAR10 shape) against the oracle's F.interpolate"""
    from naturaldiffusion_amd.inception import InceptionEngine, flatten_state_dict
    eng = InceptionEngine(flatten_state_dict(params), max_batch=2, in_hw=(64, 48), device=dev)
    g = torch.Generator().manual_seed(6)
    x = torch.rand(2, 3, 64, 48, generator=g)
    assert _rel(eng(x).cpu(), O.forward(params, x)) <= TOL
    with pytest.raises(ValueError):
        eng(torch.rand(2, 3, 32, 32))


def test_implicit_gemm_plans_against_the_im2col_plan(params, dev):
    """natinf_set_inception_conv: 0 = the im2col + GEMM plan of rounds 3-4 (bf16 activations, filters as two bf16 terms); 1 = the same arithmetic with every
    convolution as one k_conv_ring launch (csrc/conv_ring.h: taps fetched from the activation tensor, padding taps from a page of zeros, 48 / 80 channels padded to
    64 / 96 by their producer) -- the same products summed in the same order: the same features; 2 (default) = k_conv_ring on half-precision activations and ONE
    half-precision filter term per row-scaled filter: closer to the fp32 oracle than the bf16 plans.  Ragged row tiles (B = 3: 3 x 35 x 35 rows etc.) and a
    non-square input included."""
    from naturaldiffusion_amd._lib import lib, check
    from naturaldiffusion_amd.inception import InceptionEngine, flatten_state_dict
    flat = flatten_state_dict(params)
    g = torch.Generator().manual_seed(21)
    for hw, B in (((32, 32), 3), ((40, 24), 2)):
        u8 = torch.randint(0, 256, (B, hw[0], hw[1], 3), generator=g, dtype=torch.uint8)
        ref = O.forward(params, u8.permute(0, 3, 1, 2).float() / 255)
        eng = {}
        try:
            for mode in (0, 1, 2):
                check(lib.natinf_set_inception_conv(mode), "set")
                eng[mode] = InceptionEngine(flat, max_batch=4, in_hw=hw, device=dev)
        finally:
            check(lib.natinf_set_inception_conv(2), "set")
        f = {m: e(u8).cpu() for m, e in eng.items()}
        errs = {m: _rel(f[m], ref) for m in f}
        print("plans vs oracle (max-rel):", errs, " implicit bf16 vs im2col:", _rel(f[1], f[0]))
        assert all(torch.isfinite(v).all() for v in f.values())
        assert _rel(f[1], f[0]) <= 1e-4                                            # same products, same order (observed: the same bytes)
        assert all(e <= TOL for e in errs.values())
        assert errs[2] <= errs[0]                                                  # eleven significant bits against eight
        for m in (1, 2):
            assert torch.equal(eng[m](u8[1:2]).cpu(), f[m][1:2])                   # batch independence on the new plans
    assert lib.natinf_set_inception_conv(3) != 0                                  # unknown mode: refused


CONV_CASES = [   # (B, H, W, cin, cout, kh, kw, stride, ph, pw): the engine's geometries and ragged relatives
    (2, 35, 35, 48, 64, 5, 5, 1, 2, 2),       # 48 channels (padded to 64), 5 x 5, padding 2
    (3, 17, 17, 160, 192, 1, 7, 1, 0, 3),     # 1 x 7
    (3, 17, 17, 128, 128, 7, 1, 1, 3, 0),     # 7 x 1
    (2, 35, 35, 288, 384, 3, 3, 2, 0, 0),     # stride 2, no padding
    (1, 9, 13, 32, 40, 3, 3, 1, 1, 1),        # non-square, N = 40: ragged column tile
    (5, 8, 8, 448, 384, 3, 3, 1, 1, 1),       # long K
    (2, 21, 19, 96, 96, 3, 3, 2, 1, 1),       # stride 2 WITH padding (not in the network)
    (4, 12, 12, 64, 200, 1, 1, 1, 0, 0),      # 1 x 1, N = 200
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("f16", [0, 1])
def test_conv_ring_against_torch_conv2d(case, f16, dev):
    """k_conv_ring on its own (natinf_debug_conv_ring), every tile shape, bf16 with two filter terms and half precision with one + column scales, against
    torch.nn.functional.conv2d in fp32 on the SAME 16-bit operands: what is left is the fp32 summation order and the 16-bit rounding of the output."""
    from naturaldiffusion_amd._lib import lib, check, ptr, stream_ptr
    B, H, W, cin, cout, kh, kw, stride, ph, pw = case
    dt = torch.float16 if f16 else torch.bfloat16
    g = torch.Generator().manual_seed(H * 131 + cin)
    cin_p = (cin + 31) // 32 * 32
    x = torch.zeros(B, H, W, cin_p)
    x[..., :cin] = torch.randn(B, H, W, cin, generator=g)
    x = x.to(dt)
    w = torch.randn(cout, cin, kh, kw, generator=g) * (1.0 / (cin * kh * kw) ** 0.5)
    bias = torch.randn(cout, generator=g) * 0.2
    # pack: [cout][(ky*kw + kx)*cin_p + c]; bf16: hi | lo terms; half: one term, rows scaled by a power of two
    wk = torch.zeros(cout, kh, kw, cin_p)
    wk[..., :cin] = w.permute(0, 2, 3, 1)
    wk = wk.reshape(cout, kh * kw * cin_p)
    if f16:
        scale = torch.exp2(torch.floor(torch.log2(wk.abs().amax(dim=1).clamp_min(1e-30))) + 1)
        packed = (wk / scale[:, None]).to(dt)
        w_eff = packed.float() * scale[:, None]
        passes, col_scale = 1, scale.float().cuda()
    else:
        hi = wk.to(dt); lo = (wk - hi.float()).to(dt)
        packed = torch.cat([hi, lo], dim=1)
        w_eff = hi.float() + lo.float()
        passes, col_scale = 2, None
    w_eff4 = w_eff.reshape(cout, kh, kw, cin_p)[..., :cin].permute(0, 3, 1, 2).contiguous()
    ref = torch.relu(torch.nn.functional.conv2d(x.float()[..., :cin].permute(0, 3, 1, 2).double(), w_eff4.double(), bias.double(), stride=stride, padding=(ph, pw)))
    ref = ref.permute(0, 2, 3, 1).float()                                        # [B][Ho][Wo][cout]
    Ho, Wo = ref.shape[1], ref.shape[2]
    xd, pd, bd = x.cuda().contiguous(), packed.cuda().contiguous(), bias.cuda()
    zeros = torch.zeros(64, dtype=torch.uint8, device="cuda")
    for tile in ((0, 1, 2, 3) if f16 else (0, 1, 2)):
        out_ld = cout + 8                                                        # a channel slice of a wider buffer
        out = torch.full((B * Ho * Wo, out_ld), 7.0, dtype=dt, device="cuda")
        check(lib.natinf_debug_conv_ring(tile, f16, B, H, W, cin_p, cin_p, cout, kh, kw, stride, ph, pw, passes, ptr(xd), ptr(pd), ptr(bd),
                                         ptr(col_scale) if col_scale is not None else None, ptr(zeros), ptr(out), out_ld, stream_ptr()), "conv_ring")
        got = out[:, :cout].float().cpu().reshape(B, Ho, Wo, cout)
        tol = (2.0 ** -10 if f16 else 2.0 ** -7) * ref.abs().max().item()
        assert torch.isfinite(got).all() and (got - ref).abs().max().item() <= tol, (case, f16, tile, (got - ref).abs().max().item(), tol)
        assert (out[:, cout:].float() == 7.0).all()                              # nothing written beyond the slice
    assert lib.natinf_debug_conv_ring(3, 0, B, H, W, cin_p, cin_p, cout, kh, kw, stride, ph, pw, 2, ptr(xd), ptr(pd), None, None, ptr(zeros), ptr(out), out_ld, None) != 0


def test_features_do_not_depend_on_the_batch(params, dev):
    """calc_fid scores FID_BATCH (500) images per engine call where the reference feeds 50: an image's pool3 features are the same bytes either way (also for a
    ragged last batch), so the statistics -- and the FID -- are those of the reference's batching"""
    from naturaldiffusion_amd.inception import InceptionEngine, flatten_state_dict
    from naturaldiffusion_amd import CIFAR10NaturalInference as M
    flat = flatten_state_dict(params)
    g = torch.Generator().manual_seed(9)
    imgs = torch.randint(0, 256, (620, 32, 32, 3), dtype=torch.uint8, generator=g).to(dev)
    small, big = InceptionEngine(flat, max_batch=50, in_hw=(32, 32), device=dev), InceptionEngine(flat, max_batch=M.FID_BATCH, in_hw=(32, 32), device=dev)
    a = torch.cat([small(imgs[i:i + 50]) for i in range(0, 620, 50)])
    b = torch.cat([big(imgs[i:i + M.FID_BATCH]) for i in range(0, 620, M.FID_BATCH)])
    assert torch.equal(a, b) and M._fid_batch(small) == 50 and M._fid_batch(big) == M.FID_BATCH == 500
    assert np.array_equal(M.get_activation(imgs, small, 2048, dev), M.get_activation(imgs, big, 2048, dev))


def test_frechet_distance_on_the_gpu_equals_the_host_forms(dev):
    """frechet_distance(..., device=cuda): torch.linalg.eigh / eigvalsh in float64 on the device against the LAPACK eigh form and the scipy sqrtm form on the host,
    full-rank and rank-deficient statistics, with and without a FrechetReference (whose device-side root is cached)"""
    from naturaldiffusion_amd.fid_stats import FrechetReference, frechet_distance
    r = np.random.RandomState(11)
    for n, dim in ((900, 256), (100, 256), (3000, 512)):
        a = r.randn(n, dim) @ r.randn(dim, dim) * 0.1
        b = r.randn(n, dim) @ r.randn(dim, dim) * 0.1 + 0.2
        m1, s1, m2, s2 = a.mean(0), np.cov(a, rowvar=False), b.mean(0), np.cov(b, rowvar=False)
        host, gpu = frechet_distance(m1, s1, m2, s2), frechet_distance(m1, s1, m2, s2, device=dev)
        ref = FrechetReference(m1, s1)
        via_ref = frechet_distance(ref, None, m2, s2, device=dev)
        assert abs(gpu - host) <= 1e-9 * abs(host) + 1e-9 * np.trace(s1) and abs(via_ref - gpu) <= 1e-12 * abs(gpu) + 1e-12 * np.trace(s1), (n, dim, host, gpu, via_ref)
        assert ref.root_on(dev) is ref.root_on(dev)
        assert abs(gpu - frechet_distance(m1, s1, m2, s2, method="sqrtm")) <= 1e-6 * abs(host)


def test_frechet_distance_between_hip_and_oracle_statistics(engine, params, repo_root):
    """2,000 synthetic images (smooth random fields + noise, uint8): (mu, Sigma) of the engine's features vs (mu, Sigma) of the fp32
    oracle's, through the product's own statistics / Frechet code (fid_stats.py).  This is the FID a perfect sampler would be charged
    for running the Inception forward in bf16."""
    from naturaldiffusion_amd.fid_stats import ActivationStats, frechet_distance
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    g = torch.Generator().manual_seed(7)
    n = 2000
    low = torch.rand(n, 3, 8, 8, generator=g)
    imgs = torch.nn.functional.interpolate(low, size=(32, 32), mode="bilinear", align_corners=False) + 0.15 * torch.randn(n, 3, 32, 32, generator=g)
    u8 = (imgs.clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    st_h, st_o = ActivationStats(2048), ActivationStats(2048)
    worst = 0.0
    for i in range(0, n, 100):
        b = u8[i:i + 100]
        fo = O.forward(params, b.permute(0, 3, 1, 2).float() / 255)
        fh = engine(b).cpu()
        worst = max(worst, _rel(fh, fo))
        st_h.update(fh); st_o.update(fo)
    mu_h, cov_h = st_h.mean_cov()
    mu_o, cov_o = st_o.mean_cov()
    fd = frechet_distance(mu_h, cov_h, mu_o, cov_o)
    rec = {"images": n, "frechet_distance_hip_vs_oracle": fd, "max_rel_feature_error": worst, "mean_feature": float(mu_o.mean()),
           "trace_cov_oracle": float(np.trace(cov_o)), "mu_diff_sq": float(((mu_h - mu_o) ** 2).sum())}
    os.makedirs(repo_root / "gpurun_out", exist_ok=True)
    (repo_root / "gpurun_out" / "inception_fd.json").write_text(json.dumps(rec, indent=1))
    print("inception bf16 vs fp32:", json.dumps(rec))
    assert worst <= TOL
    assert fd <= 1e-2 * max(1.0, np.trace(cov_o) / 2048), rec            # <= 1e-2 at unit per-feature variance


def test_calc_fid_end_to_end_with_a_weights_file(dev, params, tmp_path, monkeypatch):
    """the script path (reference :73-86): a pt_inception-style state dict on disk + a statistics .npz -> calc_fid / calc_fid_sharded through
    the HIP engine; images scored against their OWN statistics give FID ~ 0, against shifted statistics the shift's |d mu|^2."""
    from naturaldiffusion_amd import CIFAR10NaturalInference as M
    wpath = tmp_path / "pt_inception-2015-12-05-6726825d.pth"
    sd = dict(params); sd["fc.weight"] = torch.zeros(4, 4)                # the checkpoint carries more than the pool3 path
    torch.save(sd, wpath)
    monkeypatch.setenv("NATINF_INCEPTION_WEIGHTS", str(wpath))
    g = torch.Generator().manual_seed(9)
    imgs = torch.randint(0, 256, (300, 32, 32, 3), generator=g, dtype=torch.uint8)
    model = M.fid_inception(dev)
    act = M.get_activation(imgs, model)
    assert act.shape == (300, 2048) and act.dtype == np.float64
    mu, sigma = act.mean(axis=0), np.cov(act, rowvar=False)
    ref = tmp_path / "mu_sigma.npz"
    np.savez(ref, mu=mu, sigma=sigma)
    assert abs(M.calc_fid(imgs, ref, dev)) < 1e-3 * np.trace(sigma)
    np.savez(ref, mu=mu + 0.01, sigma=sigma)
    fid = M.calc_fid(imgs, ref, dev, model=model)
    assert abs(fid - 2048 * 1e-4) < 1e-3 * np.trace(sigma) + 0.05
    monkeypatch.setenv("NATINF_INCEPTION_WEIGHTS", str(tmp_path / "absent.pth"))
    with pytest.raises(FileNotFoundError, match="fid: blocked"):
        M.calc_fid(imgs, ref, dev)
