"""Tile sharding across ranks (one process per GPU) and the control plane of the packed-map gather.

Tiles are independent (own seed / parameters / time), so a
batch is partitioned with NO data-path collective during synthesis.  The only
exchange step the north-star names is one gather of the finished packed maps
to a root rank.  The DATA path of that gather is in the product library:
`ocean_gather_maps` (include/ocean.h) = ncclGather x 2 on RCCL, zero-copy from
the context's map buffers, world_size-1 concurrent point-to-point xGMI streams
into the root.  This module holds what surrounds it on the host: the block
partition of the tiles, the distribution of the RCCL unique id through whatever
process group the harness already has (`exchange_unique_id`), the MAX-over-ranks
timing reduction of the bench contract, and `gather_maps`, a torch.distributed
gather of map tensors used by the CPU (gloo) tests of the rank-major layout.
"""
from __future__ import annotations

from typing import List, Optional, Tuple


def tile_shard(total_tiles: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block partition: returns (first_tile, count) owned by `rank`.

    Remainder tiles go to the lowest ranks, so counts differ by at most one and
    every tile is owned exactly once.
    """
    if total_tiles < 0 or world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad shard arguments")
    base, rem = divmod(total_tiles, world_size)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def tile_seeds(base_seed: int, first_tile: int, count: int) -> List[int]:
    """Seed of global tile i is base_seed + i (0x5EED0000 + tile_index)."""
    return [(base_seed + first_tile + i) & 0xFFFFFFFFFFFFFFFF for i in range(count)]


def gather_maps(local_maps, dst: int = 0, group=None, async_op: bool = False):
    """Gather every rank's packed maps tensor to `dst`.

    local_maps: tensor [tiles_per_rank, 2, N, N, 4] float32 (displacement map,
    then normal map, per tile), same shape on every rank.
    Returns (gathered, work): on dst `gathered` is [world, tiles_per_rank, 2, N, N, 4]
    (rank-major = global tile order for equal shards), elsewhere None.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out = None
    glist = None
    if rank == dst:
        out = torch.empty((world,) + tuple(local_maps.shape), dtype=local_maps.dtype, device=local_maps.device)
        glist = [out[i] for i in range(world)]
    work = dist.gather(local_maps, gather_list=glist, dst=dst, group=group, async_op=async_op)
    return out, work


def exchange_unique_id(W, src: int = 0, group=None) -> bytes:
    """Rank `src` creates the RCCL unique id through the C ABI (ocean_comm_unique_id) and every rank of the
    process group receives it (any backend: the id is 128 bytes of host data).  Single process: just creates it.
    `W` is the watersurfacerendering_amd package (passed in so this module stays importable without the library)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return W.comm_unique_id()
    box = [W.comm_unique_id() if dist.get_rank(group) == src else None]
    dist.broadcast_object_list(box, src=src, group=group)
    uid = box[0]
    assert isinstance(uid, (bytes, bytearray)) and len(uid) == 128
    return bytes(uid)


def max_over_ranks(seconds: float, device: Optional[str] = None, group=None) -> float:
    """MAX all-reduce of a scalar timing (bench contract: slowest rank defines the step time)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
