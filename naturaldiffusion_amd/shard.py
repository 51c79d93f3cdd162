"""Batch sharding of a generation job over the GPUs of one node.

Every image trajectory is independent, so the job is partitioned by *global image index* with no
collective on the data path: rank ``r`` of ``W`` owns indices ``r, r+W, r+2W, ...`` and walks them in
batches.  Only timing (max over ranks) and the optional final gather use ``torch.distributed`` (RCCL on
the GPUs, gloo in the CPU tests).  Replaces the reference's single-process ``nn.DataParallel``
(deps/score_sde_pytorch/models/utils.py:93)."""
from __future__ import annotations

from typing import Iterator, List

import torch


def rank_indices(sample_count: int, rank: int, world: int) -> range:
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return range(rank, sample_count, world)


def rank_batches(sample_count: int, batch: int, rank: int, world: int) -> Iterator[List[int]]:
    """Global image indices of each batch this rank generates (last batch may be ragged)."""
    idx = rank_indices(sample_count, rank, world)
    for s in range(0, len(idx), batch):
        yield list(idx[s:s + batch])


def max_over_ranks(seconds: float, device=None) -> float:
    """no process group: the value itself; an initialised group (of any size, one rank included) runs the all-reduce"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def gather_images(local_u8: torch.Tensor, local_index: torch.Tensor, sample_count: int) -> torch.Tensor:
    """Optional epilogue (FID): all-gather uint8 images and place them by global index.  Ragged shards are
    padded to the longest one; returns [sample_count, ...] on every rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out = torch.empty((sample_count,) + tuple(local_u8.shape[1:]), dtype=local_u8.dtype, device=local_u8.device)
        out[local_index] = local_u8
        return out
    world = dist.get_world_size()
    n = torch.tensor([local_u8.shape[0]], dtype=torch.int64, device=local_u8.device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n)
    m = int(max(int(v) for v in ns))
    pad_img = torch.zeros((m,) + tuple(local_u8.shape[1:]), dtype=local_u8.dtype, device=local_u8.device)
    pad_idx = torch.full((m,), -1, dtype=torch.int64, device=local_u8.device)
    pad_img[:local_u8.shape[0]] = local_u8
    pad_idx[:local_u8.shape[0]] = local_index
    imgs = [torch.empty_like(pad_img) for _ in range(world)]
    idxs = [torch.empty_like(pad_idx) for _ in range(world)]
    dist.all_gather(imgs, pad_img)
    dist.all_gather(idxs, pad_idx)
    out = torch.empty((sample_count,) + tuple(local_u8.shape[1:]), dtype=local_u8.dtype, device=local_u8.device)
    for im, ix in zip(imgs, idxs):
        keep = ix >= 0
        out[ix[keep]] = im[keep]
    return out
