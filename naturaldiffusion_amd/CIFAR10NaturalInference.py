"""Drop-in for ``src/CIFAR10NaturalInference.py`` (the Natural Inference part, lines 202-317).

Same public names, argument meaning and defaults as the reference script: ``to_pixel``, ``data_fn``,
``weighted_sum``, ``natural_inference_tx``, ``calc_fid``.  The sampling loop runs on one MI355X:
the NCSN++ forward in the HIP engine (include/natinf_ncsnpp.h) and one fused ``ni_step`` launch per
step (include/natinf.h).  Constants the reference edits in-source are keyword arguments with the
reference's values as defaults (batch 500, 50 000 samples, seed 888, step_5_weight_00.npz).
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Callable, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ptr, stream_ptr
from .coeff import load_coeff_npz, SparseRows
from .sampler import CifarNI, vp_std_f32

root_path = Path(__file__).resolve().parent.parent


def _to_pixel(x: torch.Tensor, centered: int, to_cpu: bool = True) -> torch.Tensor:
    _lib.require_gpu()
    x = x.contiguous()
    B, C, H, W = x.shape
    out = torch.empty((B, H, W, C), dtype=torch.uint8, device=x.device)
    check(lib.natinf_to_pixel_u8(ptr(x), ptr(out), B, C, H, W, centered, stream_ptr()), "natinf_to_pixel_u8")
    return out.cpu() if to_cpu else out            # (the copy to the host blocks: the batch pipeline of natural_inference_tx copies at the end)


def to_pixel(batch: torch.Tensor) -> torch.Tensor:
    """Reference :212-216: [B,C,H,W] fp32 in [0,1] -> uint8 [B,H,W,C] on the CPU."""
    return _to_pixel(batch, 0)


def to_pixel_from_centered(x: torch.Tensor) -> torch.Tensor:
    """inverse_scaler (datasets.py:32-38) + to_pixel in one launch: x in [-1,1] -> uint8 NHWC (CPU)."""
    return _to_pixel(x, 1)


@torch.no_grad()
def data_fn(score_fn: Callable, xt: torch.Tensor, t, x_coeff, eps_coeff, device) -> torch.Tensor:
    """Reference :219-230.  ``score_fn(xt, vec_t)`` is the caller's score function; the fp64 conversion
    ``(score*eps_coeff**2 + xt)/x_coeff`` runs in ``natinf_step_f64hist`` (std = -1 makes its internal
    ``-out/std`` the identity on the score)."""
    _lib.require_gpu()
    vec_t = t * torch.ones(xt.shape[0], device=device)
    score = score_fn(xt, vec_t).contiguous()
    xt = xt.contiguous()
    E = xt.numel()
    hist = torch.empty((1, E), dtype=torch.float64, device=xt.device)
    scratch = torch.empty(E, dtype=torch.float32, device=xt.device)
    check(lib.natinf_step_f64hist(ptr(xt), ptr(score), ptr(xt), ptr(hist), ptr(scratch), None, None, 0, 0.0, 0,
                                  float(x_coeff), float(eps_coeff), -1.0, 0.0, E, stream_ptr()), "natinf_step_f64hist")
    return hist.view(xt.shape)


def weighted_sum(past_x0_coeff: Sequence[float], seq_x0: Sequence[torch.Tensor]) -> torch.Tensor:
    """Reference :233-238: fp64 sum_i seq_x0[i]*coeff[i] (ascending i) -> fp32."""
    _lib.require_gpu()
    slab = torch.stack([s.contiguous().reshape(-1) for s in seq_x0]).to(torch.float64)
    n, E = slab.shape
    rows = SparseRows(np.asarray(past_x0_coeff, np.float64)[None, :n], lambda k: n, torch.float64, slab.device,
                      dense=True, diag=False)
    out = torch.empty(E, dtype=torch.float32, device=slab.device)
    idx, val, nt = rows.ptrs(0)
    check(lib.natinf_weighted_sum_f64(ptr(slab), ptr(out), idx, val, nt, E, stream_ptr()), "natinf_weighted_sum_f64")
    return out.view(seq_x0[0].shape)


@torch.no_grad()
def natural_inference(model_fn: Callable, noise: torch.Tensor, weight_path, dense: bool = False,
                      fast_f32: bool = False, return_all: bool = False, stds=None):
    """The loop body of ``natural_inference_tx`` (:292-304) for one batch of initial noise."""
    C, B, node = load_coeff_npz(weight_path)
    ni = CifarNI(C, B, node, noise.numel(), device=noise.device, dense=dense, fast_f32=fast_f32, stds=stds)
    return ni.run(model_fn, noise, return_all=return_all)


def philox_noise(indices, shape_per_image, seed: int, device="cuda:0") -> torch.Tensor:
    """[len(indices), *shape_per_image] fp32 N(0,1), image i keyed by its GLOBAL index (include/natinf.h,
    natinf_randn_philox_f32): identical for any GPU count / batch split."""
    _lib.require_gpu()
    idx = torch.as_tensor(list(indices), dtype=torch.int64, device=device)
    per = int(np.prod(shape_per_image))
    out = torch.empty((idx.numel(),) + tuple(shape_per_image), dtype=torch.float32, device=device)
    check(lib.natinf_randn_philox_f32(ptr(out), idx.numel(), per, ptr(idx), 0, 0, int(seed) & (2 ** 64 - 1), stream_ptr()),
          "natinf_randn_philox_f32")
    return out


class BatchLanes:
    """The batch pipeline of ``natural_inference_tx`` / ``generate_sharded``: the batches of a generation job are independent
    trajectories (reference loop :287-309), so consecutive batches go to ``len(models)`` lanes -- one HIP stream, one denoiser handle and
    one set of history slabs each -- and the under-occupied launches of one batch run under the other's convolutions.
    A batch is enqueued end to end (noise, 15-18 forwards + ni_step launches, ``to_pixel``) without the host waiting for the GPU; the uint8
    images stay on the device until ``finish``; at most ``depth`` batches per lane are enqueued ahead of the GPU."""

    def __init__(self, models, C, B, node, device, depth: int = 2):
        _lib.require_gpu()
        self.device = torch.device(device)
        self.coeff = (C, B, node)
        self.models = list(models)
        multi = len(self.models) > 1
        self.streams = [torch.cuda.Stream(device=self.device) if multi else None for _ in self.models]
        self.samplers = [dict() for _ in self.models]                   # per lane: batch size -> CifarNI (a ragged last batch gets its own slabs)
        self.pending = [[] for _ in self.models]
        self.depth = int(depth)
        self.count = 0
        self.out = []
        self.joined = [False] * len(self.models)                       # lane k has been ordered behind the caller's stream

    def _sampler(self, k: int, n: int) -> CifarNI:
        if n not in self.samplers[k]:
            self.samplers[k][n] = CifarNI(*self.coeff, n * 3 * 32 * 32, device=self.device)
        return self.samplers[k][n]

    def submit(self, n: int, noise=None, noise_fn=None) -> torch.Tensor:
        """One batch of ``n`` images: ``noise`` (drawn on the CALLER's current stream) or ``noise_fn()`` (called on the lane's stream).
        Returns the uint8 [n, 32, 32, 3] DEVICE tensor the lane will fill."""
        k = self.count % len(self.models)
        self.count += 1
        st = self.streams[k]
        if len(self.pending[k]) >= self.depth:                          # bound the host's run-ahead (and the noise tensors in flight)
            self.pending[k].pop(0).synchronize()
        with torch.cuda.device(self.device):
            # A lane's first launch is ordered behind everything the caller's stream has queued: the engine's workspace and packed weights (and
            # the blocks clone() got from the caching allocator, which may be recycled ones with work pending) were allocated THERE, and a
            # still-queued forward of lane 0's engine on the caller's stream would otherwise race with the lane's.  With `noise` every submit waits.
            if st is not None and (noise is not None or not self.joined[k]):
                st.wait_stream(torch.cuda.current_stream())
                self.joined[k] = True
            if st is not None and noise is not None:
                noise.record_stream(st)
            with torch.cuda.stream(st):
                ni = self._sampler(k, n)                                # (first use allocates on the lane's stream)
                z = noise if noise is not None else noise_fn()
                pix = _to_pixel(ni.run(self.models[k], z), 1, to_cpu=False)
                ev = torch.cuda.Event()
                ev.record()
        self.pending[k].append(ev)
        self.out.append(pix)
        return pix

    def finish(self):
        """Wait for every lane (the caller's stream then sees the images); returns the list of per-batch uint8 device tensors."""
        with torch.cuda.device(self.device):
            main = torch.cuda.current_stream()
            for st in self.streams:
                if st is not None:
                    main.wait_stream(st)
            if any(st is not None for st in self.streams):
                for pix in self.out:                                     # allocated on a lane's stream, consumed (and freed) on the caller's: the allocator must
                    pix.record_stream(main)                              # not hand the block back to the lane while the caller's torch.cat is still queued
        self.pending = [[] for _ in self.models]
        self.joined = [False] * len(self.models)
        out, self.out = self.out, []
        return out


def _lane_models(model_fn, streams: int, n_batches: int):
    """One denoiser per lane.  An ``NCSNppEngine`` is cloned (own launch plan + workspace, shared packed weights); a sequence of callables
    is taken as given (one lane each); any other callable owns state we cannot duplicate, so it gets one lane."""
    from .ncsnpp import NCSNppEngine
    if isinstance(model_fn, (list, tuple)):
        return list(model_fn)
    n = max(1, min(int(streams), n_batches))
    if isinstance(model_fn, NCSNppEngine):
        return [model_fn] + [model_fn.clone() for _ in range(n - 1)]
    return [model_fn]


@torch.no_grad()
def generate_sharded(model_fn, weight_path, sample_count: int, batch_size: int, rank: int = 0, world: int = 1,
                     seed: int = 888, device="cuda:0", streams: int = 3, to_cpu: bool = True, coeff=None):
    """Batch-sharded generation (BASELINE config 3): this rank generates the images whose
    global index is rank, rank+world, ... in batches of ``batch_size`` -- no collective on the data path -- on the two-lane pipeline
    of ``natural_inference_tx`` (``BatchLanes``): Philox noise keyed by the GLOBAL image index drawn on the lane's own stream, uint8 images
    kept on the device, one copy to the host at the end (none with ``to_cpu=False``: ``calc_fid_sharded`` scores device tensors).
    ``model_fn``: an ``NCSNppEngine`` (cloned per lane), a sequence of callables (one lane each) or one callable (one lane).
    ``coeff``: (C, B, node_coeff) instead of a file (matrices from ``coeffgen``).
    Returns (uint8 images [n_local, 32, 32, 3], their global indices [n_local] int64 on the CPU)."""
    from .shard import rank_batches
    C, B, node = coeff if coeff is not None else load_coeff_npz(weight_path)
    batches = list(rank_batches(sample_count, batch_size, rank, world))
    dev = torch.device(device)
    if not batches:
        return torch.empty((0, 32, 32, 3), dtype=torch.uint8, device="cpu" if to_cpu else dev), torch.empty(0, dtype=torch.int64)
    lanes = BatchLanes(_lane_models(model_fn, streams, len(batches)), C, B, node, dev)
    for batch in batches:
        lanes.submit(len(batch), noise_fn=lambda b=batch: philox_noise(b, (3, 32, 32), seed, dev))
    imgs = torch.cat(lanes.finish())
    idxs = torch.cat([torch.tensor(b, dtype=torch.int64) for b in batches])
    return (imgs.cpu() if to_cpu else imgs), idxs


INCEPTION_WEIGHTS = "pt_inception-2015-12-05-6726825d.pth"      # what pytorch_fid downloads on first use (FID_WEIGHTS_URL)


def inception_weights_path() -> Path:
    """$NATINF_INCEPTION_WEIGHTS, else weights/pt_inception-2015-12-05-6726825d.pth next to the coefficient files, else the torch hub
    cache pytorch_fid fills (~/.cache/torch/hub/checkpoints)."""
    env = os.environ.get("NATINF_INCEPTION_WEIGHTS")
    if env:
        return Path(env)
    for cand in (root_path / "weights" / INCEPTION_WEIGHTS, Path.home() / ".cache/torch/hub/checkpoints" / INCEPTION_WEIGHTS):
        if cand.exists():
            return cand
    return root_path / "weights" / INCEPTION_WEIGHTS


FID_BATCH = 500      # images per Inception call.  The reference feeds 50 (:47); an image's features do not depend on its batch (the same bytes at 50 and 500,
                     # tests/test_gpu_inception.py) and the engine runs 8,500 images/s in 50s, 11,400 in 500s


def _fid_batch(model) -> int:
    return max(1, min(FID_BATCH, int(getattr(model, "max_batch", 50))))


def fid_inception(device, image_hw=(32, 32), max_batch: int = FID_BATCH):
    """The pool3 network of ``calc_fid`` on the HIP library (include/natinf_inception.h) -- the reference builds
    ``InceptionV3([BLOCK_INDEX_BY_DIM[2048]])`` here (:75-77).  The weights are a download the image does not hold: without the file
    this raises (``fid: blocked``); there is no torch-module fallback."""
    from .inception import InceptionEngine, load_fid_inception_weights
    path = inception_weights_path()
    if not path.exists():
        raise FileNotFoundError(f"fid: blocked -- Inception weights {path} missing (pytorch_fid's {INCEPTION_WEIGHTS}; set NATINF_INCEPTION_WEIGHTS)")
    return InceptionEngine(load_fid_inception_weights(path), max_batch=max_batch, in_hw=image_hw, device=device)


def get_activation(imgs, model, dims: int = 2048, device=None) -> np.ndarray:
    """Reference :44-70: uint8 [n, H, W, 3] images -> pool3 activations [n, dims], ``FID_BATCH`` (or the engine's ``max_batch``) images per call
    -- the reference's 50 give the same features -- (the /255, the NCHW permutation, the 299 x 299 resize and the 2x - 1 scaling happen inside the
    engine's stem kernel)."""
    assert dims == 2048, "the HIP engine implements the pool3 (2048-dimensional) output"
    pred = np.empty((len(imgs), dims))
    bs = _fid_batch(model)
    for ii in range(0, len(imgs), bs):
        pred[ii:ii + bs] = model(imgs[ii:ii + bs]).cpu().numpy()
    return pred


def calc_fid(imgs, ref_path, device, model=None):
    """Reference :73-86 (InceptionV3 pool3 + Frechet distance).  Needs the Inception weights and the ``cifar10_mu_sigma.npz``
    statistics, neither of which ships with the reference."""
    from .fid_stats import frechet_distance
    ref = _ref_statistics(ref_path)
    on_gpu = torch.device(device).type == "cuda"
    if not on_gpu:
        ref.prefetch()                                                  # (host path: the reference covariance's square root is taken while the images go through Inception)
    model = model or fid_inception(device, tuple(imgs.shape[1:3]))
    act = get_activation(imgs, model, 2048, device)
    mu, sigma = np.mean(act, axis=0), np.cov(act, rowvar=False)
    return frechet_distance(ref, None, mu, sigma, device=device if on_gpu else None)


_REF_CACHE = {}


def _ref_statistics(ref):
    """``ref``: path of a ``cifar10_mu_sigma.npz``-style file (keys mu, sigma), a (mu, sigma) pair, or a ``fid_stats.FrechetReference`` -> a FrechetReference
    (unpacks as (mu, sigma); a file's object is kept per path and modification time, so every FID of a job shares the square root of the reference covariance)."""
    from .fid_stats import FrechetReference
    if isinstance(ref, FrechetReference):
        return ref
    if isinstance(ref, (tuple, list)):
        return FrechetReference(ref[0], ref[1])
    if not os.path.exists(ref):
        raise FileNotFoundError(f"fid: blocked -- {ref} (CIFAR10 Inception statistics) is missing")
    key = (os.path.abspath(str(ref)), os.path.getmtime(ref))
    if key not in _REF_CACHE:
        f = np.load(ref)
        _REF_CACHE.clear()
        _REF_CACHE[key] = FrechetReference(f["mu"], f["sigma"])
    return _REF_CACHE[key]


class PendingFid:
    """A Frechet distance being evaluated on a host thread (``calc_fid_sharded(..., defer=True)``): ``result()`` waits for it and returns the FID
    (None on the ranks that do not evaluate it); ``seconds`` is the thread's own wall time once it is done."""

    def __init__(self, fn):
        import threading
        self._value, self._error, self.seconds = None, None, 0.0

        def run():
            import time
            t0 = time.perf_counter()
            try:
                self._value = fn()
            except BaseException as e:                                   # surfaced by result()
                self._error = e
            self.seconds = time.perf_counter() - t0
        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()

    def result(self):
        self._thread.join()
        if self._error is not None:
            raise self._error
        return self._value


def calc_fid_sharded(imgs, ref_path, device, group=None, model=None, timings: Optional[dict] = None, root_only: bool = False, defer: bool = False):
    """``calc_fid`` for a batch-sharded run: every rank scores ITS images (uint8 [n_local, H, W, 3], on the device or the host; ``FID_BATCH`` or the
    engine's ``max_batch`` per Inception call -- the reference's 50 give the same features), the (count, sum, outer-product sum) statistics are summed over ranks with ONE all-reduce (33.6 MB of fp64,
    fid_stats.ActivationStats) instead of gathering images or activations, and every rank returns the same FID.  ``timings`` (optional
    dict) receives the wall seconds of the three parts: inception_s, allreduce_s, frechet_s.  ``root_only``: only rank 0 evaluates the
    Frechet distance (a 2048 x 2048 matrix square root on the host: eight ranks doing it at once fight for the same cores), the others
    return None.  ``defer``: the host part (two symmetric eigen-decompositions, ~1.7 s on 8 cores; the reference side's is shared by every FID against the
    same statistics and starts when this function is entered) runs on a thread and a ``PendingFid`` is returned -- a job that scores several image sets
    (config 3 scores two coefficient matrices) generates the next set on the GPU meanwhile.  (The Inception PASS beside the next generation too -- the whole scoring
    on a thread and a HIP stream of its own -- was built and measured: both are bound by the same compute units; generation 4.67 -> 5.27 s, the job's 7.05-7.17 s
    unchanged.  Not kept.)"""
    import time
    from .fid_stats import ActivationStats, frechet_distance
    import torch.distributed as dist
    is_root = not (dist.is_available() and dist.is_initialized()) or dist.get_rank(group) == 0
    ref = _ref_statistics(ref_path)
    fdev = device if torch.device(device).type == "cuda" else None       # the Frechet distance's eigen-decompositions on the GPU (0.1 s) or, on a CPU device, on the host (1.4-2 s)
    if (is_root or not root_only) and fdev is None:
        ref.prefetch()
    model = model or fid_inception(device, tuple(imgs.shape[1:3]))
    dev = torch.device(device)
    sync = (lambda: torch.cuda.current_stream(dev).synchronize()) if dev.type == "cuda" else (lambda: None)      # (this call's stream, not the device)
    t0 = time.perf_counter()
    st = None
    bs = _fid_batch(model)
    for i in range(0, len(imgs), bs):
        a = model(imgs[i:i + bs])
        if st is None:
            st = ActivationStats(a.shape[-1], device=device)             # (pool3: 2048)
        st.update(a)
    if st is None:                                                       # a rank without images still takes part in the all-reduce
        st = ActivationStats(2048, device=device)
    sync()
    t1 = time.perf_counter()
    st.all_reduce(group)
    sync()
    t2 = time.perf_counter()
    fid = None
    if is_root or not root_only:
        mu, cov = st.mean_cov()
        if defer:
            fid = PendingFid(lambda: frechet_distance(ref, None, mu, cov, device=fdev))
        else:
            fid = frechet_distance(ref, None, mu, cov, device=fdev)
    elif defer:
        fid = PendingFid(lambda: None)
    if timings is not None:
        timings.update(inception_s=t1 - t0, allreduce_s=t2 - t1, frechet_s=time.perf_counter() - t2, images_all_ranks=int(float(st.n)))
    return fid


class FidBlocked:
    """What ``natural_inference_tx(compute_fid=True)`` returns when the FID assets (Inception weights, ``cifar10_mu_sigma.npz``) are absent:
    the images it generated and the reason -- distinguishable from both a FID value (float) and the ``compute_fid=False`` result (a tensor)."""

    def __init__(self, images: torch.Tensor, reason: str):
        self.images, self.reason = images, reason

    def __repr__(self):
        return f"FidBlocked({tuple(self.images.shape)}, {self.reason!r})"


@torch.no_grad()
def natural_inference_tx(batch_size: int = 500,
                         ckpt_filename: Optional[str] = None,
                         weight_path: Optional[str] = None,
                         sample_count: int = 50 * 1000, seed: int = 888, device: str = "cuda:0",
                         compute_fid: bool = True, flat_params: Optional[torch.Tensor] = None, streams: int = 3):
    """Reference :242-317: generate ``sample_count`` CIFAR10 images with the NI matrix at ``weight_path``
    and score them.  ``flat_params`` lets a caller supply weights directly (engine order) instead of the
    score_sde checkpoint.

    ``streams`` (default 3 since round 6; 2 before): the batches are independent trajectories (reference loop :287-309); consecutive batches rotate over that many HIP streams
    (one engine handle + history buffer each), so that the under-occupied launches of one batch -- the 4x4 level, the per-sample GroupNorm
    tables, every launch's last round of blocks -- run under the other batch's convolutions: -6 % per batch at 512 images on one MI355X.
    The noise is drawn in the reference's order and every batch runs the same launches: the images are bit-identical to ``streams=1``,
    the reference's one-after-the-other order (tests/test_gpu_ncsnpp.py; DESIGN.md section 5 for what that took)."""
    from .ncsnpp import NCSNppEngine, load_score_sde_checkpoint
    ckpt_filename = ckpt_filename or str(root_path / "deps/score_sde_pytorch/checkpoint_8.pth")
    weight_path = weight_path or str(root_path / "weights/step_5_weight_00.npz")
    if flat_params is None:
        assert os.path.exists(ckpt_filename)
        flat_params = load_score_sde_checkpoint(ckpt_filename)
    C, B, node = load_coeff_npz(weight_path)
    print(C / np.diag(C)[:, None])
    print(weight_path)
    bz = batch_size
    num = int(np.ceil(sample_count / bz))
    engine = NCSNppEngine(flat_params, max_batch=batch_size, device=device)
    lanes = BatchLanes(_lane_models(engine, streams, num), C, B, node, device)       # the second lane shares the first's packed weights
    torch.cuda.synchronize(torch.device(device))
    torch.manual_seed(seed)
    for ii in range(num):
        print("processing", ii)
        noise = torch.randn(bz, 3, 32, 32, dtype=torch.float32, device=device)      # the reference's stream of normals, in its order (:290)
        lanes.submit(bz, noise=noise)
    all_batch = torch.concatenate(lanes.finish()).cpu()                                # ONE copy to the host, after the last batch
    if not compute_fid:
        return all_batch
    try:
        fid_value = calc_fid(all_batch, root_path / "weights/cifar10_mu_sigma.npz", device)
    except FileNotFoundError as e:                          # the assets are downloads: say so, keep the images
        print(e)
        return FidBlocked(all_batch, str(e))
    print(fid_value)
    print(weight_path)
    print(C / np.diag(C)[:, None])
    return fid_value


if __name__ == "__main__":
    natural_inference_tx()
