"""GPU parity tests of the NCSN++ HIP engine against the fp32 oracle (itself pinned to the reference's
nn.Module by tests/test_oracle_ncsnpp.py).  bf16 operands / fp32 accumulate vs an fp32 reference:
tolerance = 3e-2 of the max magnitude on the network output (SURVEY section 7 proposes <= 3e-2 for bf16
paths; observed 1.75e-2..1.8e-2) and a PER-LEVEL bound on every intermediate module output (TOL_BY_LEVEL below:
about 1.2x the largest error observed at that level over the B = 2 and B = 512 plans -- 4e-3 at the stem rising to
3.5e-2 in the 4x4 blocks, whose outputs are small differences of large sums, then falling again); the observed
errors are printed and written to gpurun_out/ for the record."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ncsnpp_oracle as N
from oracle import ni_oracle as O

TOL = 3e-2          # final output
# (first module, last module, bound, observed max over rounds 2-3 at B = 2 / B = 512): the module table is ncsnpp.module_table()
TOL_BY_LEVEL = [(2, 2, 6e-3, "stem 4.3e-3"), (3, 7, 1.2e-2, "down 32x32: 9.4e-3"), (8, 16, 1.8e-2, "down 16x16 + attention: 1.47e-2 (B = 2); 1.68e-2 at B = 512 (module 14).  Round 5 had widened this to 2.2e-2 for k_attn_blk256 (1.89e-2: the GroupNorm partial "
                 "sums of the attention output were added up in another token order); round 6 adds them up in the two-launch order again (attn_blk256.h, dpp_row_sum_tau) and the bound is back"),
                (17, 21, 2.2e-2, "down 8x8: 1.8e-2"), (22, 34, 3.8e-2, "4x4 level, middle and 4x4 up blocks: 3.33e-2 (B = 2, module 27) / 3.48e-2 (B = 512, module 29)"),
                (35, 40, 2.9e-2, "up 8x8: 2.46e-2"), (41, 47, 2.7e-2, "up 16x16: 2.25e-2"), (48, 52, 1.9e-2, "up 32x32: 1.5e-2")]


def tol_module(k: int) -> float:
    return next(t for a, b, t, _ in TOL_BY_LEVEL if a <= k <= b)


def first_module_over_its_bound(report):
    return next((k for k in range(2, 53) if report[f"tap{k:02d}"] > tol_module(k)), None)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def params():
    return N.make_params(seed=0)


@pytest.fixture(scope="module")
def flat(params):
    from naturaldiffusion_amd.ncsnpp import flatten_state_dict
    return flatten_state_dict(params)


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


def test_forward_per_module(dev, params, flat, golden_dir, repo_root):
    from naturaldiffusion_amd.ncsnpp import NCSNppEngine
    fx = np.load(golden_dir / "ncsnpp_forward.npz")
    x, labels = torch.from_numpy(fx["x"]), torch.from_numpy(fx["labels"])
    taps = {}
    y_ref = N.forward(params, x, labels, taps)
    assert np.abs(y_ref.numpy() - fx["y"]).max() <= 2e-5 * np.abs(fx["y"]).max()       # oracle == reference module
    eng = NCSNppEngine(flat, max_batch=2, device=dev, keep_activations=True)
    y = eng(x.to(dev), labels.to(dev))
    torch.cuda.synchronize()
    report = {}
    worst = (0.0, None)
    for k in range(2, 53):
        got = eng.tap(k, tuple(taps[k].shape)).cpu()
        e = _rel(got, taps[k])
        report[f"tap{k:02d}"] = e
        if e > worst[0]:
            worst = (e, k)
    report["y"] = _rel(y.cpu(), y_ref)
    os.makedirs(repo_root / "gpurun_out", exist_ok=True)
    (repo_root / "gpurun_out" / "ncsnpp_tap_errors.json").write_text(json.dumps(report, indent=1))
    print("per-module max-rel errors:", json.dumps(report))
    assert torch.isfinite(y).all()
    first_bad = first_module_over_its_bound(report)
    assert first_bad is None, f"module {first_bad} first exceeds its level's bound {tol_module(first_bad)}: {report[f'tap{first_bad:02d}']:.3e}"
    assert report["y"] <= TOL, report["y"]


def test_batch_independence_and_reuse_mode(dev, flat, golden_dir):
    """the arena-reusing plan (production) gives the same numbers as the keep-all plan, and a sample's
    output does not depend on its batch neighbours or on the batch size (ragged last tile included)."""
    from naturaldiffusion_amd.ncsnpp import NCSNppEngine
    fx = np.load(golden_dir / "ncsnpp_forward.npz")
    g = torch.Generator().manual_seed(5)
    x = torch.cat([torch.from_numpy(fx["x"]), torch.randn(5, 3, 32, 32, generator=g)]).to(dev)
    labels = torch.cat([torch.from_numpy(fx["labels"]), torch.rand(5, generator=g) * 999]).to(dev)
    keep = NCSNppEngine(flat, max_batch=2, device=dev, keep_activations=True)
    prod = NCSNppEngine(flat, max_batch=7, device=dev)
    y2 = keep(x[:2], labels[:2]).clone()
    y7 = prod(x, labels).clone()
    y3 = prod(x[:3], labels[:3]).clone()
    torch.cuda.synchronize()
    assert torch.equal(y7[:2], y2)
    assert torch.equal(y7[:3], y3)
    # determinism: same launch twice
    assert torch.equal(prod(x, labels), y7)


def test_ni_end_to_end_with_engine(dev, params, flat, repo_root):
    """BASELINE config 1 (step_5_weight_00.npz) at B=2: HIP engine + ni_step vs the all-CPU oracle path."""
    from naturaldiffusion_amd.ncsnpp import NCSNppEngine
    from naturaldiffusion_amd.CIFAR10NaturalInference import natural_inference
    C, B, node = O.load_coeff_npz(repo_root / "weights/step_5_weight_00.npz")
    g = torch.Generator().manual_seed(888)
    noise = torch.randn(2, 3, 32, 32, generator=g)
    ref = O.cifar_ni_trajectory(N.model_fn_from_params(params), noise, C, B, node)
    eng = NCSNppEngine(flat, max_batch=2, device=dev)
    xs = natural_inference(eng, noise.to(dev), repo_root / "weights/step_5_weight_00.npz", return_all=True)
    errs = [_rel(a.cpu(), b) for a, b in zip(xs[1:], ref[1:])]
    print("trajectory max-rel errors per step:", errs)
    assert max(errs) <= 5e-2          # bf16 denoiser error propagated through 5 NI steps (|C| rows sum up to 3.8)


def test_every_gemm_variant_gives_the_same_network(dev, flat, golden_dir):
    """force each DMA/ring tile variant for all eligible launches: outputs agree to bf16 rounding noise (the
    variants only differ in tiling / pipeline depth; accumulation order inside a K loop is identical)."""
    from naturaldiffusion_amd.ncsnpp import NCSNppEngine
    from naturaldiffusion_amd._lib import lib
    fx = np.load(golden_dir / "ncsnpp_forward.npz")
    x, labels = torch.from_numpy(fx["x"]).to(dev), torch.from_numpy(fx["labels"]).to(dev)
    eng = NCSNppEngine(flat, max_batch=2, device=dev)
    outs = {}
    try:
        # 26 / 27 (dma256x256h / dma512x128h) are the automatic choices of the big launches at B = 512, 9 / 17 / 8 of the rest;
        # the superseded pipelines (2-7, 10-16, 18-22, 24, 25) exist in -DNATINF_DEV builds only and are refused here
        for v in (2, 3, 4, 5, 6, 7, 10, 11, 12, 13, 14, 15, 16, 18, 19, 20, 21, 22, 24, 25):
            assert lib.natinf_set_gemm_variant(v) != 0, v
        for v in (0, 1, 8, 9, 17, 26, 27):
            assert lib.natinf_set_gemm_variant(v) == 0
            outs[v] = eng(x, labels).clone()
            torch.cuda.synchronize()
    finally:
        lib.natinf_set_gemm_variant(0)
    ref = torch.from_numpy(fx["y"])
    for v, y in outs.items():
        assert _rel(y.cpu(), ref) <= TOL, (v, _rel(y.cpu(), ref))
        assert torch.equal(y, outs[0]) or _rel(y.cpu(), outs[0].cpu()) < 2e-2


def _describe_gemms(eng, B):
    import ctypes as C
    from naturaldiffusion_amd._lib import lib
    buf = C.create_string_buffer(1 << 16)
    n = lib.natinf_ncsnpp_describe_gemms(eng._h, B, buf, len(buf))
    assert n > 0
    return [ln.split() for ln in buf.value.decode().strip().split("\n")]


def test_bench_batch_512_runs_the_benchmarked_kernels_and_matches_golden(dev, flat, golden_dir):
    """BASELINE config 2's batch: at B = 512 the dispatcher picks the tile variants bench.py is timed on (the LDS-DMA
    tiles are never chosen at B <= 7; k_conv_gn runs 2048 / 1024 blocks instead of 8 / 4).  The two golden samples sit in slots 0-1 AND 510-511 of a batch of noise: both pairs must
    match the reference module's output (fixture y) within TOL and be bit-identical to each other (a sample's result
    depends neither on its slot nor on its neighbours)."""
    from naturaldiffusion_amd.ncsnpp import NCSNppEngine
    fx = np.load(golden_dir / "ncsnpp_forward.npz")
    gx, gl = torch.from_numpy(fx["x"]), torch.from_numpy(fx["labels"])
    g = torch.Generator().manual_seed(11)
    x = torch.randn(512, 3, 32, 32, generator=g)
    labels = torch.rand(512, generator=g) * 999
    for s in (0, 510):
        x[s:s + 2] = gx
        labels[s:s + 2] = gl
    eng = NCSNppEngine(flat, max_batch=512, device=dev)
    chosen = {}
    for f in _describe_gemms(eng, 512):
        chosen.setdefault(f[6].split("/")[0], []).append((int(f[0]), int(f[1])))
    # the families that carry the bench's device time: the fused GroupNorm+SiLU convolution (every res-block conv of the 32x32 and
    # 16x16 levels but Conv_0 of the down-sampling blocks: 41 launches) and the fused output head (k_head_conv).  The hand-pipelined 512x128 /
    # 256x256 LDS-DMA tiles (variants 27 / 26) are under test in test_every_gemm_variant_gives_the_same_network and carry the VAE / SD3 engines
    assert len(chosen.get("conv_gn", [])) >= 40, {k: len(v) for k, v in chosen.items()}
    assert len(chosen.get("head_conv", [])) == 1, {k: len(v) for k, v in chosen.items()}      # (round 3: the 128 -> 3 head left the 512x128 tile for k_head_conv)
    y = eng(x.to(dev), labels.to(dev))
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    ref = torch.from_numpy(fx["y"])
    head, tail = y[0:2].cpu(), y[510:512].cpu()
    assert _rel(head, ref) <= TOL, _rel(head, ref)
    assert _rel(tail, ref) <= TOL, _rel(tail, ref)
    assert torch.equal(head, tail)
    # and the same samples through the B = 2 plan (other tile variants) agree to bf16 rounding noise
    y2 = NCSNppEngine(flat, max_batch=2, device=dev)(gx.to(dev), gl.to(dev)).cpu()
    assert _rel(head, y2) < 2e-2


def test_per_module_taps_at_the_benchmarked_batch(dev, params, flat, golden_dir, repo_root):
    """Round-2 review, weak #2: the 51 per-module taps at B = 512 -- where k_conv_gn2 launches thousands of blocks and the dispatcher picks
    the LDS-DMA tiles bench.py is timed on -- not only at B = 2.  The two golden samples ride in slots 0-1 and 510-511 of a batch of
    random neighbours on a keep_activations plan (28.7 MB per image: 14.7 GB); every module output of both pairs is held to the same
    per-level bound (TOL_BY_LEVEL) as the B = 2 test, and the two pairs are bit-identical to each other at every module."""
    from naturaldiffusion_amd.ncsnpp import NCSNppEngine
    fx = np.load(golden_dir / "ncsnpp_forward.npz")
    gx, gl = torch.from_numpy(fx["x"]), torch.from_numpy(fx["labels"])
    taps = {}
    y_ref = N.forward(params, gx, gl, taps)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(512, 3, 32, 32, generator=g)
    labels = torch.rand(512, generator=g) * 999
    for s in (0, 510):
        x[s:s + 2] = gx
        labels[s:s + 2] = gl
    eng = NCSNppEngine(flat, max_batch=512, device=dev, keep_activations=True)
    y = eng(x.to(dev), labels.to(dev))
    torch.cuda.synchronize()
    report = {}
    for k in range(2, 53):
        full = eng.tap(k, (512,) + tuple(taps[k].shape[1:]))
        head, tail = full[0:2].cpu(), full[510:512].cpu()
        del full
        assert torch.equal(head, tail), k
        report[f"tap{k:02d}"] = _rel(head, taps[k])
    report["y"] = _rel(y[0:2].cpu(), y_ref)
    os.makedirs(repo_root / "gpurun_out", exist_ok=True)
    (repo_root / "gpurun_out" / "ncsnpp_tap_errors_b512.json").write_text(json.dumps(report, indent=1))
    print("per-module max-rel errors at B = 512:", json.dumps(report))
    first_bad = first_module_over_its_bound(report)
    assert first_bad is None, f"module {first_bad} first exceeds its level's bound {tol_module(first_bad)}: {report[f'tap{first_bad:02d}']:.3e}"
    assert report["y"] <= TOL, report["y"]


def test_fused_and_split_k_plans_against_the_unfused_plan(dev, flat, golden_dir):
    """Plan-build-time switches (round-2 advisor note): the default plan -- GroupNorm + SiLU inside the 3x3 convolutions, up-sampling inside
    their fetches, split-K on the 4x4 level (it engages at this batch: 16 tiles of 128 x 128 for 256 CUs) -- against the plans with each
    of those turned off, on the same 64 samples: every pair agrees to bf16 rounding noise, and the golden samples stay within TOL."""
    from naturaldiffusion_amd.ncsnpp import NCSNppEngine
    from naturaldiffusion_amd._lib import lib
    fx = np.load(golden_dir / "ncsnpp_forward.npz")
    g = torch.Generator().manual_seed(21)
    x = torch.randn(64, 3, 32, 32, generator=g)
    labels = torch.rand(64, generator=g) * 999
    x[:2] = torch.from_numpy(fx["x"]); labels[:2] = torch.from_numpy(fx["labels"])
    xd, ld = x.to(dev), labels.to(dev)
    base_eng = NCSNppEngine(flat, max_batch=64, device=dev)
    assert any("splitk" in r[6] for r in _describe_gemms(base_eng, 64))
    base = base_eng(xd, ld).clone()
    ref = torch.from_numpy(fx["y"])
    assert _rel(base[:2].cpu(), ref) <= TOL
    outs = {}
    assert any(r[6].startswith("head_conv") for r in _describe_gemms(base_eng, 64))
    for name, setter in (("fuse_gn", lib.natinf_set_fuse_gn), ("fuse_up", lib.natinf_set_fuse_up), ("fuse_head", lib.natinf_set_fuse_head),
                         ("fuse_gn8", lib.natinf_set_fuse_gn8), ("fuse_gn4", lib.natinf_set_fuse_gn4), ("fuse_fin", lib.natinf_set_fuse_fin), ("attn_proj", lib.natinf_set_attn_proj), ("attn_qkv", lib.natinf_set_attn_qkv)):
        try:
            assert setter(0) == 0
            eng = NCSNppEngine(flat, max_batch=64, device=dev)          # the switch is read when the plan is built
        finally:
            setter(1)
        rows = _describe_gemms(eng, 64)
        if name == "fuse_gn":
            assert not any(r[6].startswith("conv_gn") for r in rows)
        if name == "fuse_head":
            assert not any(r[6].startswith("head_conv") for r in rows)
        if name == "fuse_gn8":                                       # without it the 8x8 level is back on the LDS-DMA implicit GEMM
            n8 = lambda rr: sum(1 for r in rr if r[6].startswith("conv_gn") and int(r[0]) == 64 * 64)
            assert n8(rows) == 0 and n8(_describe_gemms(base_eng, 64)) >= 18
        if name == "fuse_gn4":                                       # without it the 4x4 level is back on k_gn_apply + split-K GEMM + reduce + k_gn_stats
            n4 = lambda rr: sum(1 for r in rr if r[6].startswith("conv_gn") and int(r[0]) == 64 * 16)
            assert n4(rows) == 0 and n4(_describe_gemms(base_eng, 64)) >= 20
        outs[name] = eng(xd, ld).clone()
    try:
        assert lib.natinf_set_gemm_splitk(0) == 0
        assert not any("splitk" in r[6] for r in _describe_gemms(base_eng, 64))
        outs["splitk"] = base_eng(xd, ld).clone()
    finally:
        lib.natinf_set_gemm_splitk(1)
    torch.cuda.synchronize()
    for name, y in outs.items():
        assert torch.isfinite(y).all()
        assert _rel(y.cpu(), base.cpu()) < 2e-2, (name, _rel(y.cpu(), base.cpu()))
        assert _rel(y[:2].cpu(), ref) <= TOL, name
    assert lib.natinf_set_conv_gn8_tile(0) != 0 and lib.natinf_set_conv_gn8_tile(1) == 0      # the two-image tile of the 8x8 level: development builds only
    # the fp32-slab A/B knob no longer breaks the fused plan (round-2 advisor, medium): the fused convolutions ignore it
    try:
        assert lib.natinf_set_gemm_epilogue(1) == 0
        y = base_eng(xd, ld)
        torch.cuda.synchronize()
    finally:
        lib.natinf_set_gemm_epilogue(0)
    assert _rel(y.cpu(), base.cpu()) < 2e-2


def test_workspace_too_small_is_an_error(dev, flat):
    from naturaldiffusion_amd.ncsnpp import NCSNppEngine
    from naturaldiffusion_amd._lib import lib, ptr, stream_ptr
    eng = NCSNppEngine(flat, max_batch=2, device=dev)
    x = torch.zeros(4, 3, 32, 32, device=dev)
    with pytest.raises(ValueError):
        eng(x, torch.zeros(4, device=dev))
    out = torch.empty_like(x)
    rc = lib.natinf_ncsnpp_forward(eng._h, ptr(x), ptr(torch.zeros(4, device=dev)), ptr(out), 4, ptr(eng._ws), eng._ws.numel(), stream_ptr())
    assert rc == -1


@pytest.mark.parametrize("B", [1, 3, 5, 7, 13])
def test_odd_batches_default_plan_against_the_unfused_plan(dev, flat, B):
    """Tails of everything that packs several samples into a tile or pairs blocks: the 4x4 level's four images per tile (batch % 4), the attention
    kernels' XCD pairing (2 B % 16), producer-written GroupNorm tables of a partial tile -- the default plan against the plan with those fusions off."""
    from naturaldiffusion_amd.ncsnpp import NCSNppEngine
    from naturaldiffusion_amd._lib import lib
    g = torch.Generator().manual_seed(100 + B)
    x = torch.randn(B, 3, 32, 32, generator=g).to(dev)
    labels = (torch.rand(B, generator=g) * 999).to(dev)
    y = NCSNppEngine(flat, max_batch=B, device=dev)(x, labels).clone()
    knobs = [lib.natinf_set_fuse_gn8, lib.natinf_set_fuse_gn4, lib.natinf_set_fuse_fin, lib.natinf_set_attn_qkv, lib.natinf_set_attn_proj]
    try:
        for k in knobs:
            assert k(0) == 0
        ref = NCSNppEngine(flat, max_batch=B, device=dev)(x, labels).clone()
    finally:
        for k in knobs:
            k(1)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    assert _rel(y.cpu(), ref.cpu()) < 2e-2


@pytest.mark.parametrize("B", [2, 67])
def test_producer_written_tables_at_16x16_are_the_finalize_kernels_bytes(dev, flat, B):
    """natinf_set_fuse_fin: 1 (default) = at 16x16 k_conv_gn3's 256 x 256 tile -- ONE sample, every channel -- writes its consumer's GroupNorm table in its
    epilogue; 2 = the round-4 plan (a k_gn_finalize launch per 16x16 table).  Same sums in the same order: the network's output is the same bytes.  Then the
    run-time fallback: with natinf_set_conv_gn_w128(0) every 16x16 launch takes k_conv_gn2, whose tiles cannot write the table -- the plan that claimed
    producer-written tables must launch k_gn_finalize itself and still give the bytes of the plan that never claimed them."""
    from naturaldiffusion_amd.ncsnpp import NCSNppEngine
    from naturaldiffusion_amd._lib import lib
    g = torch.Generator().manual_seed(300 + B)
    x = torch.randn(B, 3, 32, 32, generator=g).to(dev)
    labels = (torch.rand(B, generator=g) * 999).to(dev)
    new = NCSNppEngine(flat, max_batch=B, device=dev)
    try:
        assert lib.natinf_set_fuse_fin(2) == 0
        old = NCSNppEngine(flat, max_batch=B, device=dev)
    finally:
        lib.natinf_set_fuse_fin(1)
    y_new, y_old = new(x, labels).clone(), old(x, labels).clone()
    assert torch.isfinite(y_new).all() and torch.equal(y_new, y_old)
    assert torch.equal(new(x, labels), y_new)                                  # (the tables are rewritten every forward)
    try:
        assert lib.natinf_set_conv_gn_w128(0) == 0                             # k_conv_gn2 everywhere: no launch writes a 16x16 table
        z_new, z_old = new(x, labels).clone(), old(x, labels).clone()
    finally:
        lib.natinf_set_conv_gn_w128(7)
    # (against y_new only close: k_conv_gn2's 128-row tiles give TWO partial rows per 16x16 sample, so the statistics are summed in another order)
    assert torch.equal(z_new, z_old) and _rel(z_new.cpu(), y_new.cpu()) < 2e-2
    assert torch.equal(new(x, labels), y_new)                                  # back on k_conv_gn3: the producer writes again
    assert lib.natinf_set_fuse_fin(4) != 0


def test_natural_inference_tx_two_stream_pipeline(dev, flat, repo_root):
    """CIFAR10NaturalInference.natural_inference_tx(streams=2, the default): consecutive batches on two HIP streams (two engine handles) against the
    reference's one-after-the-other order.  Same noise order, same launches per batch: bit-identical images, run after run (it was not, until the DPP
    reduction of the GroupNorm partial sums stopped reading packed-fp32 results: DESIGN.md section 5).  5 batches of 8: the streams end unevenly."""
    from naturaldiffusion_amd import CIFAR10NaturalInference as M
    w = str(repo_root / "weights" / "step_5_weight_00.npz")
    run = lambda s: M.natural_inference_tx(batch_size=8, weight_path=w, sample_count=40, seed=7, device=dev, compute_fid=False, flat_params=flat, streams=s)
    a, b = run(1), run(2)
    assert a.shape == (40, 32, 32, 3) and a.dtype == torch.uint8
    for _ in range(3):
        assert torch.equal(run(2), a)
    assert torch.equal(b, a)


def test_two_engines_on_two_streams_are_bit_reproducible(dev, flat):
    """Two engine handles running concurrently on two HIP streams, 64 images each: every forward equals the one computed alone.  (The failure this guards
    against needed a wave of another kernel on the same SIMD: 30-40 of 40 such forwards differed in a whole image's GroupNorm statistics.)"""
    from naturaldiffusion_amd.ncsnpp import NCSNppEngine
    B = 64
    e1, e2 = NCSNppEngine(flat, max_batch=B, device=dev), NCSNppEngine(flat, max_batch=B, device=dev)
    g = torch.Generator(device="cpu").manual_seed(11)
    x1, x2 = torch.randn(B, 3, 32, 32, generator=g).to(dev), torch.randn(B, 3, 32, 32, generator=g).to(dev)
    t = (torch.rand(B, generator=g) * 999).to(dev)
    r1, r2 = e1(x1, t).clone(), e2(x2, t).clone()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    bad = 0
    for _ in range(20):
        with torch.cuda.stream(s2):
            o2 = e2(x2, t)
        with torch.cuda.stream(s1):
            o1 = e1(x1, t)
        torch.cuda.synchronize()
        bad += int(not torch.equal(o1, r1)) + int(not torch.equal(o2, r2))
    assert bad == 0, bad
