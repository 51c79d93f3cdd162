"""BASELINE config 3 as a workload (reference src/CIFAR10NaturalInference.py:281-312): batch-sharded generation on the two-lane
pipeline + the Inception pool3 engine + ``calc_fid_sharded``'s one all-reduce of statistics, against ``calc_fid`` on the gathered
images -- at world size 1 in-process and at world size 2 with two gloo ranks sharing cuda:0 (RCCL refuses two ranks on one
device; the collective's code path is the same ``dist.all_reduce``).  Synthetic NCSN++ / Inception weights and (0, I) reference
statistics stand in for the downloads; what is checked is the plumbing: shares, ragged last batch, lanes, statistics, Frechet."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
N_TOTAL, BATCH = 210, 64                     # two ranks: 105 images each = one full batch of 64 + a ragged batch of 41


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _engines(dev):
    from naturaldiffusion_amd.inception import InceptionEngine
    from naturaldiffusion_amd.ncsnpp import NCSNppEngine
    from naturaldiffusion_amd.synth import synthetic_flat_params, synthetic_inception_flat
    return NCSNppEngine(synthetic_flat_params(0), max_batch=BATCH, device=dev), InceptionEngine(synthetic_inception_flat(0), max_batch=50, device=dev)


def _coeff():
    from naturaldiffusion_amd.coeff import load_coeff_npz
    return load_coeff_npz(ROOT / "weights" / "step_5_weight_00.npz")


def test_engine_clone_shares_weights_and_is_bit_identical(dev):
    from naturaldiffusion_amd._lib import lib
    eng, _ = _engines(dev)
    twin = eng.clone()
    assert twin._packed.data_ptr() == eng._packed.data_ptr() and twin._ws.data_ptr() != eng._ws.data_ptr()
    g = torch.Generator().manual_seed(1)
    x, lab = torch.randn(BATCH, 3, 32, 32, generator=g).to(dev), (torch.rand(BATCH, generator=g) * 999).to(dev)
    assert torch.equal(eng(x, lab), twin(x, lab))
    import ctypes as C
    h = C.c_void_p()
    assert lib.natinf_ncsnpp_create(C.byref(h), 2) == 0                     # a `ddpm` plan cannot adopt NCSN++ weights
    assert lib.natinf_ncsnpp_share(h, eng._h) != 0 and lib.natinf_ncsnpp_share(eng._h, eng._h) != 0
    h2 = C.c_void_p()
    assert lib.natinf_ncsnpp_create(C.byref(h2), 0) == 0
    assert lib.natinf_ncsnpp_share(twin._h, h2) != 0                        # nothing loaded into h2 yet
    lib.natinf_ncsnpp_destroy(h); lib.natinf_ncsnpp_destroy(h2)


def test_generate_sharded_lanes_and_calc_fid_sharded_world1(dev):
    """two lanes == one lane bit for bit (incl. the ragged last batch); images stay on the device; calc_fid_sharded == calc_fid"""
    from naturaldiffusion_amd import CIFAR10NaturalInference as M
    eng, inc = _engines(dev)
    co = _coeff()
    two, i2 = M.generate_sharded(eng, None, N_TOTAL, BATCH, device=dev, streams=2, to_cpu=False, coeff=co)
    one, i1 = M.generate_sharded(eng, None, N_TOTAL, BATCH, device=dev, streams=1, to_cpu=True, coeff=co)
    assert two.is_cuda and not one.is_cuda and two.dtype == torch.uint8 and tuple(two.shape) == (N_TOTAL, 32, 32, 3)
    assert torch.equal(i1, torch.arange(N_TOTAL)) and torch.equal(i1, i2) and torch.equal(two.cpu(), one)
    ref = (np.zeros(2048), np.eye(2048))
    tm = {}
    a = M.calc_fid_sharded(two, ref, dev, model=inc, timings=tm)
    b = M.calc_fid(one, ref, dev, model=inc)
    assert np.isfinite(a) and abs(a - b) <= 1e-5 * abs(b) + 1e-6, (a, b)
    assert tm["images_all_ranks"] == N_TOTAL and tm["allreduce_s"] < 0.5 and tm["inception_s"] > 0


def test_sharding_invariance_with_the_real_engine(dev):
    """An image depends on (seed, its GLOBAL index) only -- also through the bf16 engine, as long as the batches have one size (the launch plan's tile choices
    depend on the batch size, a sample's arithmetic does not depend on its neighbours: GroupNorm is per sample): 128 images as one rank's 4 batches of 32 and as
    four ranks' single batches of 32 give the same bytes per global index."""
    from naturaldiffusion_amd import CIFAR10NaturalInference as M
    eng, _ = _engines(dev)
    co = _coeff()
    one, i1 = M.generate_sharded(eng, None, 128, 32, device=dev, coeff=co)
    full = torch.empty_like(one)
    for r in range(4):
        im, ix = M.generate_sharded(eng, None, 128, 32, rank=r, world=4, device=dev, coeff=co)
        assert torch.equal(ix, torch.arange(r, 128, 4)) and im.shape[0] == 32
        full[ix] = im
    assert torch.equal(i1, torch.arange(128)) and torch.equal(full, one)


_RANK_SCRIPT = r"""
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
from naturaldiffusion_amd import CIFAR10NaturalInference as M
from naturaldiffusion_amd.coeff import load_coeff_npz
from naturaldiffusion_amd.inception import InceptionEngine
from naturaldiffusion_amd.ncsnpp import NCSNppEngine
from naturaldiffusion_amd.shard import gather_images
from naturaldiffusion_amd.synth import synthetic_flat_params, synthetic_inception_flat
dev = torch.device("cuda:0")
eng = NCSNppEngine(synthetic_flat_params(0), max_batch={batch}, device=dev)
inc = InceptionEngine(synthetic_inception_flat(0), max_batch=50, device=dev)
co = load_coeff_npz({root!r} + "/weights/step_5_weight_00.npz")
imgs, idx = M.generate_sharded(eng, None, {n}, {batch}, rank=rank, world=world, device=dev, to_cpu=False, coeff=co)
ref = (np.zeros(2048), np.eye(2048))
fid = M.calc_fid_sharded(imgs, ref, dev, model=inc)
full = gather_images(imgs.cpu(), idx, {n})                # gloo: host tensors
out = dict(rank=rank, n_local=int(imgs.shape[0]), fid_sharded=fid)
if rank == 0:
    out["fid_gathered"] = M.calc_fid(full, ref, dev, model=inc)
    out["idx_ok"] = bool(torch.equal(idx, torch.arange(rank, {n}, world)))
print("RESULT " + json.dumps(out), flush=True)
dist.destroy_process_group()
"""


def test_calc_fid_sharded_two_ranks_equals_calc_fid_on_the_gathered_images(dev, tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT.format(root=str(ROOT), n=N_TOTAL, batch=BATCH))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29653",
                        str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    res = [json.loads(ln.split("RESULT ", 1)[1]) for ln in p.stdout.splitlines() if "RESULT " in ln]
    assert len(res) == 2
    r0 = next(r for r in res if r["rank"] == 0); r1 = next(r for r in res if r["rank"] == 1)
    assert r0["n_local"] == 105 and r1["n_local"] == 105 and r0["idx_ok"]
    assert r0["fid_sharded"] == r1["fid_sharded"]                                  # every rank holds the same reduced statistics
    assert abs(r0["fid_sharded"] - r0["fid_gathered"]) <= 1e-5 * abs(r0["fid_gathered"]) + 1e-6, r0


def _bench(extra, timeout=1500):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "fid50k", "--no-cpu-baseline"] + extra, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_fid50k_one_command(dev):
    """`python3 bench.py --workload fid50k` at a reduced job size: rank 0's share of a 4-way job of 2,000 images = 500 = 3 full batches of 128 + 116"""
    line = _bench(["--fid-samples", "2000", "--fid-share-of", "4", "--batch", "128"])
    assert line["images"] == 500 and line["share_of"] == 4 and line["batches"] == 4 and line["last_batch"] == 116
    assert line["fid"] == "blocked" and set(line["frechet_vs_synthetic_ref"]) == {"dpmsolverpp2s_018", "ddim_vp_018"}
    assert all(line["s"][k] >= 0 for k in ("gen", "inception", "allreduce", "frechet")) and line["value"] > 0
    assert np.isfinite(line["d_matrices"]) and line["config"]["nfe"] == 18


def test_bench_fid50k_two_ranks_on_one_gpu(dev):
    line = _bench(["--gpus", "2", "--backend", "gloo", "--same-device", "--fid-samples", "1000", "--batch", "128"])
    assert line["n_gpus"] == 2 and line["share_of"] == 2 and line["images"] == 500 and line["scaling"] == "strong"
    assert line["s"]["allreduce"] > 0                                               # the statistics did cross ranks
