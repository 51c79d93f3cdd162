"""CPU-side checks: the C-ABI library loads and exports every symbol the headers declare, the engine's
static plan agrees with the oracle's description of the network, and the host logic around the kernels
(sparse coefficient rows, schedules) is right.  No compute call is made (there is no GPU here)."""
import ctypes as C
import re

import numpy as np
import pytest
import torch

from oracle import ncsnpp_oracle as N
from oracle import ni_oracle as O


def _declared(header_text):
    return set(re.findall(r"\b(natinf_[a-z0-9_]+)\s*\(", header_text))


def test_library_exports_every_declared_symbol(repo_root):
    from naturaldiffusion_amd import _lib
    names = set()
    for h in sorted(p.name for p in (repo_root / "include").glob("natinf*.h")):           # every public header
        names |= _declared((repo_root / "include" / h).read_text())
    names -= {"natinf_stream_t", "natinf_ncsnpp_t", "natinf_dit_t", "natinf_mmdit_t", "natinf_vae_t", "natinf_inception_t"}
    assert len(names) >= 20
    for n in sorted(names):
        assert hasattr(_lib.lib, n), f"libnatinf.so does not export {n}"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) <= names
    assert _lib.lib.natinf_abi_version() == 1
    assert _lib.lib.natinf_strerror(-1) == b"invalid argument"


def test_product_does_not_import_the_oracle(repo_root):
    for p in (repo_root / "naturaldiffusion_amd").rglob("*.py"):
        src = p.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f"{p} imports the oracle"
    for p in (repo_root / "naturaldiffusion_amd" / "csrc").glob("*"):
        if p.is_file():
            assert "oracle" not in p.read_text(errors="ignore").lower() or p.name == "Makefile"


def test_engine_plan_matches_oracle_plan():
    from naturaldiffusion_amd import ncsnpp
    from naturaldiffusion_amd._lib import lib
    tab = ncsnpp.module_table()
    ref = N.plan()
    assert len(tab) == len(ref) == 55
    for row, m in zip(tab, ref):
        idx, kind, cin, cout, up, down, res, poff = row
        assert (idx, kind, cin, cout, bool(up), bool(down), res) == (m.idx, m.kind, m.cin, m.cout, m.up, m.down, m.res)
    shapes = N.param_shapes()
    lay = ncsnpp.param_layout()
    assert [n for n, _ in lay] == list(shapes.keys())
    assert all(tuple(s) == tuple(shapes[n]) for n, s in lay)
    total = sum(int(np.prod(s)) for _, s in lay)
    assert total == lib.natinf_ncsnpp_param_count() == 61804419
    # module parameter offsets are the running sum in that order
    offs = {}
    run = 0
    for n, s in lay:
        i = int(n.split(".")[1])
        offs.setdefault(i, run)
        run += int(np.prod(s))
    assert all(offs[row[0]] == row[7] for row in tab)


def test_workspace_queries():
    from naturaldiffusion_amd._lib import lib
    h = C.c_void_p()
    assert lib.natinf_ncsnpp_create(C.byref(h), 0) == 0
    w1, w512 = lib.natinf_ncsnpp_workspace_bytes(h, 1), lib.natinf_ncsnpp_workspace_bytes(h, 512)
    assert w512 == 512 * w1 and 1e6 < w1 < 2e7
    hk = C.c_void_p()
    assert lib.natinf_ncsnpp_create(C.byref(hk), 1) == 0
    assert lib.natinf_ncsnpp_workspace_bytes(hk, 1) > w1
    assert lib.natinf_ncsnpp_create(C.byref(C.c_void_p()), 4) == -1                  # flags: 1 = keep activations, 2 = the `ddpm` network
    # forward before load -> ESTATE; bad args -> EINVAL (no launch happens)
    assert lib.natinf_ncsnpp_forward(h, 1, 1, 1, 1, 1, 1 << 40, None) == -4
    assert lib.natinf_ncsnpp_forward(h, None, None, None, 1, None, 0, None) == -1
    assert lib.natinf_ncsnpp_destroy(h) == 0 and lib.natinf_ncsnpp_destroy(hk) == 0


def test_flatten_state_dict_and_ema_order():
    from naturaldiffusion_amd import ncsnpp
    shapes = N.param_shapes()
    sd = {"module." + k: torch.full(s, float(i)) for i, (k, s) in enumerate(shapes.items())}
    flat = ncsnpp.flatten_state_dict(sd)
    ema = ncsnpp.flatten_ema([torch.full(s, float(i)) for i, s in enumerate(shapes.values())])
    assert flat.numel() == 61804419 and torch.equal(flat, ema)
    with pytest.raises(ValueError):
        ncsnpp.flatten_state_dict({**sd, "module.all_modules.2.weight": torch.zeros(128, 3, 1, 1)})
    with pytest.raises(KeyError):
        ncsnpp.flatten_state_dict({k: v for k, v in sd.items() if not k.endswith("all_modules.54.bias")})


def test_sparse_rows(repo_root):
    from naturaldiffusion_amd.coeff import SparseRows, load_coeff_npz, load_sd3_csv
    C_, B_, node = load_coeff_npz(repo_root / "weights/step_15_weight_173.npz")
    for dense in (False, True):
        rows = SparseRows(C_, lambda k: k + 1, torch.float64, None, dense=dense)
        for k, r in enumerate(rows.rows):
            idx = rows.idx_host[r.start:r.start + r.n]
            val = rows.val_host[r.start:r.start + r.n]
            assert np.all(np.diff(idx) > 0) and (r.n == 0 or idx.max() < k)
            assert r.diag == C_[k, k]
            rebuilt = np.zeros(k + 1)
            rebuilt[idx] = val
            rebuilt[k] = r.diag
            assert np.array_equal(rebuilt, C_[k, :k + 1])
            assert r.n == (k if dense else np.count_nonzero(C_[k, :k]))
    W = load_sd3_csv(repo_root / "weights/sd3_step_28_weight.csv")
    rows = SparseRows(W, lambda k: k + 1, torch.float32, None)
    assert rows.val_host.dtype == np.float32
    tot = 0
    for j in range(28):
        tot = tot + W[27][j]
    assert rows.rows[27].total == float(tot)
    with pytest.raises(ValueError):
        load_coeff_npz(repo_root / "tests/golden/cifar_form.npz")


def test_host_schedule_matches_oracle(repo_root):
    from naturaldiffusion_amd.sampler import vp_std_f32
    from naturaldiffusion_amd.coeff import load_coeff_npz
    _, _, node = load_coeff_npz(repo_root / "weights/step_10_weight_42.npz")
    for t in node[:-1, 0]:
        assert vp_std_f32(t) == float(O.vp_std_f32(t))


def test_gpu_required_is_loud():
    from naturaldiffusion_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.require_gpu()
    from naturaldiffusion_amd.sampler import CifarNI
    with pytest.raises(RuntimeError):
        CifarNI(np.eye(2), np.ones((2, 2)), np.ones((3, 3)), 8)


def test_synthetic_weights_match_the_oracle_recipe():
    """bench.py feeds the GPU engine from naturaldiffusion_amd.synth and its CPU baseline from the oracle:
    the two generators must produce the same tensors."""
    from naturaldiffusion_amd.synth import synthetic_state_dict
    a, b = synthetic_state_dict(0), N.make_params(0)
    assert list(a) == list(b)
    assert all(torch.equal(a[k], b[k]) for k in a)


def test_synthetic_dit_weights_match_the_oracle_recipe():
    from oracle import dit_oracle as D
    from naturaldiffusion_amd.synth import synthetic_dit_state_dict
    from naturaldiffusion_amd.dit import param_layout
    a, b = synthetic_dit_state_dict(2, 128, seed=7), D.make_params(2, 128, seed=7)
    assert list(a) == list(b) == [n for n, _ in param_layout(2, 128)]
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_synthetic_mmdit_weights_match_the_oracle_recipe():
    from oracle import mmdit_oracle as M
    from naturaldiffusion_amd.synth import synthetic_mmdit_flat
    from naturaldiffusion_amd.mmdit import flatten_state_dict
    cfg = dict(layers=2, heads=2, joint_dim=64, pooled_dim=32)
    a = synthetic_mmdit_flat(grid=8, seed=5, pos_max=24, pos_base=8, **cfg)
    b = flatten_state_dict(M.make_params(seed=5, pos_max=24, pos_base=8, **cfg), 8, **cfg)
    assert torch.equal(a, b)


def test_vae_loader_accepts_pre_rename_attention_keys():
    """The 2022 sd-vae-ft-* weight files keep the mid-block attention under query / key / value / proj_attn (some as 1x1
    conv weights); AutoencoderKL.from_pretrained renames them on load (reference ValidateNaturalInference.py:212-214 goes
    through it), so the raw-file loader must as well."""
    import torch
    from naturaldiffusion_amd.vae import flatten_state_dict, param_layout
    g = torch.Generator().manual_seed(2)
    new = {"decoder." + n: torch.randn(s, generator=g) for n, s in param_layout(4)}
    new["post_quant_conv.weight"] = torch.randn(4, 4, 1, 1, generator=g)
    new["post_quant_conv.bias"] = torch.randn(4, generator=g)
    old = {}
    for k, v in new.items():
        for a, b in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
            if f".attentions.0.{a}." in k:
                k = k.replace(f".attentions.0.{a}.", f".attentions.0.{b}.")
                if k.endswith("weight") and b in ("query", "proj_attn"):
                    v = v[:, :, None, None]                 # conv-style storage of the same matrix
        old[k] = v
    assert any(".query." in k for k in old) and not any(".to_q." in k for k in old)
    assert torch.equal(flatten_state_dict(old, 4, prefix="decoder."), flatten_state_dict(new, 4, prefix="decoder."))


def test_attention_abi_rejects_padding_beyond_the_masked_tile():
    """natinf_attention_hd64_bf16 masks only the last 128-key tile: Tp - T >= 128 must be an argument error, not a wrong
    softmax (include/natinf_mmdit.h)."""
    from naturaldiffusion_amd._lib import lib
    dummy = 4096
    assert lib.natinf_attention_hd64_bf16(dummy, dummy, 64, 256 * 64, dummy, dummy, 64, 256 * 64, 1, 1, 256, 128, 0.125, None) == -1
    assert lib.natinf_attention_hd64_bf16(dummy, dummy, 64, 256 * 64, dummy, dummy, 64, 256 * 64, 1, 1, 256, 300, 0.125, None) == -1


def test_validation_grid_is_one_row_of_eight(tmp_path):
    """reference ValidateNaturalInference.py:236: save_image(samples, path, nrow=8, ...): 8 images -> 1 x 8."""
    import torch
    from PIL import Image
    from naturaldiffusion_amd.ValidateNaturalInference import save_image_grid
    save_image_grid(torch.zeros(8, 3, 16, 16), tmp_path / "g.png")
    assert Image.open(tmp_path / "g.png").size == (8 * 18 + 2, 18 + 2)


def test_png_writer_gives_the_pixels_back(tmp_path):
    """write_png_rgb (the band-parallel deflate of round 5): a reader (PIL) decodes every file to the array it was given -- one band, several bands, ragged band
    heights, a one-pixel image -- and the stream passes PIL's integrity check (CRCs, Adler-32 of the concatenated bands)."""
    import numpy as np
    from PIL import Image
    from naturaldiffusion_amd.ValidateNaturalInference import write_png_rgb, save_image_grid
    import torch
    r = np.random.RandomState(3)
    for shape, threads in (((260, 2066, 3), 8), ((1, 1, 3), 8), ((17, 5, 3), 8), ((300, 33, 3), 3), ((64, 64, 3), 1), ((131, 7, 3), 16)):
        a = r.randint(0, 256, shape).astype(np.uint8)
        if shape[0] > 100:
            a[: shape[0] // 2] = 7                                          # a compressible half: exercises real deflate blocks next to stored ones
        write_png_rgb(a, tmp_path / "a.png", threads=threads)
        Image.open(tmp_path / "a.png").verify()
        assert np.array_equal(np.array(Image.open(tmp_path / "a.png").convert("RGB")), a), (shape, threads)
    with pytest.raises(ValueError):
        write_png_rgb(np.zeros((4, 4), np.uint8), tmp_path / "b.png")
    # through save_image_grid: the values torchvision's save_image(normalize=True, value_range=(-1, 1)) would write
    x = torch.linspace(-1.2, 1.2, 8 * 3 * 16 * 16).reshape(8, 3, 16, 16)
    save_image_grid(x, tmp_path / "g.png")
    got = np.array(Image.open(tmp_path / "g.png").convert("RGB"))
    want = (((x.clamp(-1, 1) + 1) * 0.5) * 255 + 0.5).clamp(0, 255).to(torch.uint8)
    assert np.array_equal(got[2:18, 2:18], want[0].permute(1, 2, 0).numpy()) and np.array_equal(got[2:18, 20:36], want[1].permute(1, 2, 0).numpy())
    assert (got[:2] == 0).all() and (got[:, :2] == 0).all()


def test_generation_pipeline_host_logic(tmp_path):
    """Round 4 host pieces that need no GPU: how `generate_sharded` / `natural_inference_tx` pick their lanes, the reference-statistics argument of the FID
    functions, the FidBlocked result, the handle-sharing ABI's argument checks, the per-workload defaults of bench.py."""
    import ctypes as C
    from naturaldiffusion_amd import CIFAR10NaturalInference as M
    from naturaldiffusion_amd._lib import lib
    f, g = (lambda x, t: x), (lambda x, t: -x)
    assert M._lane_models([f, g], 2, 10) == [f, g]                         # a sequence: one lane each, as given
    assert M._lane_models(f, 2, 10) == [f]                                  # an opaque callable owns state we cannot duplicate: one lane
    mu, sg = np.zeros(4), np.eye(4)
    a, b = M._ref_statistics((mu, sg))
    assert a is not None and np.array_equal(b, sg)
    np.savez(tmp_path / "ref.npz", mu=mu + 1, sigma=2 * sg)
    a, b = M._ref_statistics(tmp_path / "ref.npz")
    assert np.array_equal(a, mu + 1) and np.array_equal(b, 2 * sg)
    with pytest.raises(FileNotFoundError, match="fid: blocked"):
        M._ref_statistics(tmp_path / "absent.npz")
    fb = M.FidBlocked(torch.zeros(3, 32, 32, 3, dtype=torch.uint8), "fid: blocked -- test")
    assert tuple(fb.images.shape) == (3, 32, 32, 3) and "blocked" in fb.reason and "FidBlocked" in repr(fb)
    # natinf_ncsnpp_share: host-side argument / state checks (no GPU involved)
    h1, h2, h3 = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.natinf_ncsnpp_create(C.byref(h1), 0) == 0 and lib.natinf_ncsnpp_create(C.byref(h2), 0) == 0 and lib.natinf_ncsnpp_create(C.byref(h3), 2) == 0
    assert lib.natinf_ncsnpp_share(h2, h1) == -4                            # NATINF_ESTATE: nothing loaded into h1
    assert lib.natinf_ncsnpp_share(h1, h1) == -1 and lib.natinf_ncsnpp_share(None, h1) == -1      # NATINF_EINVAL
    for h in (h1, h2, h3):
        lib.natinf_ncsnpp_destroy(h)
    assert lib.natinf_set_flash_mode(7) == -1 and lib.natinf_set_flash_mode(3) == 0 and lib.natinf_set_flash_mode(0) == 0 and lib.natinf_set_flash_mode(3) == 0
    assert lib.natinf_set_flash_mode(1) in (0, -4) and lib.natinf_set_flash_mode(3) == 0           # intermediate forms: -DNATINF_DEV builds only


def test_bench_line_stays_small(repo_root):
    """the default line must survive a 3 KB log tail: the committed line of the round's final run is the check (bench.py moved every explanation to its docstring)"""
    import json
    f = repo_root / "profiles" / "r04" / "final_bench.json"
    line = f.read_text().strip()
    assert len(line) <= 3072, len(line)
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"]) and set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"])
    assert list(d)[:14][-2:] == ["single_stream", "pipeline"] or "single_stream" in list(d)[:16]       # the like-for-like figure sits with the headline, in front of the sub-objects
    for k in ("sd3", "sd3_fp8", "fid50k", "validate"):
        assert "value" in d[k], k
