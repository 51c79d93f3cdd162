"""FID statistics epilogue (naturaldiffusion_amd/fid_stats.py): sharded sufficient statistics + Frechet distance."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from naturaldiffusion_amd.fid_stats import ActivationStats, frechet_distance


def test_stats_match_numpy_mean_and_cov():
    g = torch.Generator().manual_seed(0)
    a = torch.randn(500, 24, generator=g) * 3 + 1
    st = ActivationStats(24)
    for i in range(0, 500, 50):                        # the reference feeds Inception in batches of 50 (:47)
        st.update(a[i:i + 50])
    mu, cov = st.mean_cov()
    assert np.allclose(mu, np.mean(a.double().numpy(), axis=0), atol=1e-12)
    assert np.allclose(cov, np.cov(a.double().numpy(), rowvar=False), atol=1e-10)


def test_frechet_distance_closed_forms():
    mu = np.array([1.0, -2.0, 0.5])
    s = np.diag([1.0, 4.0, 0.25])
    assert abs(frechet_distance(mu, s, mu, s)) < 1e-9
    # commuting (diagonal) covariances: sum (sqrt(a) - sqrt(b))^2 + |dmu|^2
    s2 = np.diag([4.0, 1.0, 0.25])
    mu2 = mu + np.array([0.0, 3.0, 4.0])
    want = 25.0 + (1 - 2) ** 2 + (2 - 1) ** 2
    assert abs(frechet_distance(mu, s, mu2, s2) - want) < 1e-8
    # symmetric in its arguments, also for non-commuting matrices
    r = np.random.RandomState(1)
    a, b = r.randn(6, 6), r.randn(6, 6)
    A, B = a @ a.T + np.eye(6), b @ b.T + np.eye(6)
    assert abs(frechet_distance(mu[:1].repeat(6), A, mu[:1].repeat(6), B) - frechet_distance(mu[:1].repeat(6), B, mu[:1].repeat(6), A)) < 1e-8


def test_eigh_form_equals_the_sqrtm_form():
    """the default (symmetric eigen-decompositions) against the pytorch_fid form (scipy sqrtm of the product), full-rank and rank-deficient statistics"""
    r = np.random.RandomState(3)
    for n, dim in ((400, 96), (60, 96), (3000, 256)):                     # 60 samples in 96 dimensions: singular covariances
        a = r.randn(n, dim) @ r.randn(dim, dim) * 0.2
        b = r.randn(n, dim) @ r.randn(dim, dim) * 0.2 + 0.3
        s1, s2, m1, m2 = np.cov(a, rowvar=False), np.cov(b, rowvar=False), a.mean(0), b.mean(0)
        e, q = frechet_distance(m1, s1, m2, s2), frechet_distance(m1, s1, m2, s2, method="sqrtm")
        assert abs(e - q) <= 1e-6 * abs(q), (n, dim, e, q)
        assert abs(frechet_distance(m1, s1, m1, s1)) <= 1e-6 * np.trace(s1)      # (square roots of eigenvalues near zero: sqrt(eps) of the trace at best)
    with pytest.raises(ValueError):
        frechet_distance(m1, s1, m2, s2, method="cholesky")


def test_reference_object_shares_its_root_and_deferred_evaluation_gives_the_same_number(tmp_path):
    """FrechetReference: the reference side's square root is taken once (also from a background thread) and frechet_distance(ref, None, ...) equals the plain call;
    calc_fid_sharded(defer=True) hands back a PendingFid whose result() is the undeferred FID; a file's reference object is shared between calls."""
    from naturaldiffusion_amd.fid_stats import FrechetReference
    from naturaldiffusion_amd import CIFAR10NaturalInference as M
    r = np.random.RandomState(7)
    a, b = r.randn(500, 64) @ r.randn(64, 64) * 0.3, r.randn(700, 64) @ r.randn(64, 64) * 0.3 + 0.1
    m1, s1, m2, s2 = a.mean(0), np.cov(a, rowvar=False), b.mean(0), np.cov(b, rowvar=False)
    ref = FrechetReference(m1, s1).prefetch()
    mu_, sg_ = ref
    assert mu_ is ref.mu and sg_ is ref.sigma
    want = frechet_distance(m1, s1, m2, s2)
    assert abs(frechet_distance(ref, None, m2, s2) - want) <= 1e-12 * abs(want) and ref.root() is ref.root()
    assert abs(frechet_distance(ref, None, m2, s2, method="sqrtm") - want) <= 1e-6 * abs(want)          # (the sqrtm form ignores the cached root)
    # through calc_fid_sharded on the host, a stand-in for the Inception engine (features = a fixed projection of the pixels)
    np.savez(tmp_path / "ref.npz", mu=m1, sigma=s1)
    proj = torch.from_numpy(r.randn(32 * 32 * 3, 64) * 0.01)
    model = lambda im: im.reshape(im.shape[0], -1).double() @ proj
    imgs = torch.from_numpy(r.randint(0, 256, (230, 32, 32, 3)).astype(np.uint8))
    tm = {}
    now = M.calc_fid_sharded(imgs, tmp_path / "ref.npz", "cpu", model=model, timings=tm)
    later = M.calc_fid_sharded(imgs, tmp_path / "ref.npz", "cpu", model=model, defer=True)
    assert isinstance(later, M.PendingFid) and later.result() == now and later.seconds > 0 and tm["images_all_ranks"] == 230
    assert M._ref_statistics(tmp_path / "ref.npz") is M._ref_statistics(tmp_path / "ref.npz")
    bad = M.PendingFid(lambda: 1 / 0)
    with pytest.raises(ZeroDivisionError):
        bad.result()


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(5)
    acts = torch.randn(400, 16, generator=g)                       # every rank can rebuild the full set
    st = ActivationStats(16)
    st.update(acts[rank::world])                                    # rank r owns images r, r+W, ... (shard.py)
    st.all_reduce()
    mu, cov = st.mean_cov()
    if rank == 0:
        np.savez(out, mu=mu, cov=cov, full_mu=np.mean(acts.double().numpy(), 0), full_cov=np.cov(acts.double().numpy(), rowvar=False))
    dist.destroy_process_group()


def test_sharded_statistics_equal_the_unsharded_ones(tmp_path):
    out = str(tmp_path / "fid.npz")
    mp.spawn(_worker, args=(2, 29641, out), nprocs=2, join=True)
    r = np.load(out)
    assert np.allclose(r["mu"], r["full_mu"], atol=1e-12) and np.allclose(r["cov"], r["full_cov"], atol=1e-10)
