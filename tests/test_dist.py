"""Multi-process CPU tests (gloo, world_size 2) of the tile sharding and the
packed-map gather used for N>1 GPUs, plus host-logic properties."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from watersurfacerendering_amd import dist as wdist


@pytest.mark.parametrize("total,world", [(64, 8), (7, 2), (1, 4), (0, 3), (65, 8), (5, 5)])
def test_tile_shard_partitions_exactly(total, world):
    owned = []
    counts = []
    for r in range(world):
        first, count = wdist.tile_shard(total, world, r)
        owned += list(range(first, first + count))
        counts.append(count)
    assert owned == list(range(total))
    assert max(counts) - min(counts) <= 1


def test_tile_shard_rejects_bad_arguments():
    for args in ((4, 0, 0), (4, 2, 2), (4, 2, -1), (-1, 2, 0)):
        with pytest.raises(ValueError):
            wdist.tile_shard(*args)


def test_tile_seeds_follow_global_index():
    assert wdist.tile_seeds(0x5EED0000, 8, 3) == [0x5EED0008, 0x5EED0009, 0x5EED000A]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        total, n = 6, 8
        first, count = wdist.tile_shard(total, world, rank)
        # stand-in maps: value encodes (global tile, map, texel) so order mistakes show up
        local = torch.empty((count, 2, n, n, 4), dtype=torch.float32)
        for i in range(count):
            for m in range(2):
                local[i, m] = (first + i) * 1000 + m * 100 + torch.arange(n * n * 4, dtype=torch.float32).reshape(n, n, 4) * 1e-3
        gathered, _ = wdist.gather_maps(local, dst=0)
        t = wdist.max_over_ranks(1.0 + rank)
        # control plane of the native gather: rank 0 creates the RCCL id through the C ABI, every rank receives it
        import watersurfacerendering_amd as W
        uid = wdist.exchange_unique_id(W, src=0)
        ids = [None] * world
        dist.all_gather_object(ids, uid)
        assert len(uid) == 128 and all(i == uid for i in ids) and any(uid)
        if rank == 0:
            flat = gathered.reshape(total, 2, n, n, 4)
            ok = all(float(flat[g, m, 0, 0, 0]) == g * 1000 + m * 100 for g in range(total) for m in range(2))
            q.put((ok, tuple(gathered.shape), t))
        else:
            assert gathered is None
            q.put((True, None, t))
    finally:
        dist.destroy_process_group()


def test_gather_maps_gloo_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[0] for r in res)
    assert any(r[1] == (2, 3, 2, 8, 8, 4) for r in res)
    assert all(r[2] == 2.0 for r in res)       # MAX over ranks of (1.0, 2.0)


def test_max_over_ranks_single_process():
    assert wdist.max_over_ranks(0.25) == 0.25
