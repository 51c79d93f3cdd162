"""The N>1 path on CPU: world_size-2 gloo processes exercise the batch sharding, the max-over-ranks
timing reduction and the optional gather epilogue (no GPU, no data-path collective)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from naturaldiffusion_amd import shard


def test_partition_is_exact():
    for count, batch, world in ((50000, 512, 8), (10000, 500, 8), (1001, 64, 3), (7, 4, 2), (3, 8, 4)):
        seen = []
        for r in range(world):
            for b in shard.rank_batches(count, batch, r, world):
                assert 0 < len(b) <= batch
                seen += b
        assert sorted(seen) == list(range(count))
    assert len(list(shard.rank_batches(50000, 512, 0, 8))) == 13          # ceil(6250/512), SURVEY section 8d cfg 3
    with pytest.raises(ValueError):
        shard.rank_indices(10, 2, 2)


def _worker(rank, world, port, count, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        idx = torch.tensor(list(shard.rank_indices(count, rank, world)), dtype=torch.int64)
        imgs = (idx % 251).to(torch.uint8)[:, None, None, None].expand(-1, 2, 2, 3).contiguous()   # image i is filled with i % 251
        full = shard.gather_images(imgs, idx, count)
        ok = bool((full[:, 0, 0, 0] == (torch.arange(count) % 251).to(torch.uint8)).all())
        t = shard.max_over_ranks(1.0 + rank)
        q.put((rank, ok, t))
    finally:
        dist.destroy_process_group()


def test_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 37, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert [r[2] for r in res] == [2.0, 2.0]            # max over ranks, identical on every rank
