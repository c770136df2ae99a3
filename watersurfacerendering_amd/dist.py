"""Tile sharding across ranks (one process per GPU) and the packed-map gather.

Tiles are independent (own seed / parameters / time), so a
batch is partitioned with NO data-path collective during synthesis.  The only
exchange step the north-star names is one gather of the finished packed maps
to a root rank: `torch.distributed.gather` (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests).  Each peer has a direct xGMI link to the
root, so the gather is world_size-1 concurrent point-to-point streams.
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


def max_over_ranks(seconds: float, device: Optional[str] = None, group=None) -> float:
    """MAX all-reduce of a scalar timing (bench contract: slowest rank defines the step time)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
