"""Rank process of tests/test_zz_multi_gpu.py: one GPU per rank, tiles sharded, the library's own RCCL gather.
Launched by torch.distributed.run; prints GATHER_OK on rank 0 when every tile of every rank arrived intact."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dist.init_process_group("gloo")                       # control plane only: the data path is the library's RCCL communicator
    import watersurfacerendering_amd as W
    from watersurfacerendering_amd import dist as wdist
    n, tiles_total, seed = 256, 2 * world + 1, 0x5EED0000  # uneven shard on purpose: equal counts are required, so pad
    per = -(-tiles_total // world)
    first = rank * per
    b = W.OceanBatch(n, per, local)
    b.prepare(seed + first)
    b.comm_init(world, rank, wdist.exchange_unique_id(W, src=0))
    recv = torch.full((2, world, per, n, n, 4), -1.0, dtype=torch.float32, device=f"cuda:{local}") if rank == 0 else None
    rp = (recv[0].data_ptr(), recv[1].data_ptr()) if rank == 0 else (None, None)
    ok = True
    for depth in (1, 2):
        b.set_pipeline_depth(depth)
        for j in range(3):
            b.compute_waves_async(0.25 * j)
            b.gather_maps(0, *rp)
        b.synchronize()
        d, q = b.read_maps()
        mine = [torch.from_numpy(d), torch.from_numpy(q)]
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        if rank == 0:
            got = recv.cpu()
            for r in range(world):
                ok = ok and torch.equal(got[0, r], everyone[r][0]) and torch.equal(got[1, r], everyone[r][1])
            # rank-major = global tile order: tile g of the batch has seed + g
            ref = W.OceanBatch(n, 1, local); ref.prepare(seed + per * (world - 1)); ref.compute_waves(0.5)
            rd, rq = ref.read_maps(); ref.close()
            ok = ok and np.array_equal(got[0, world - 1, 0].numpy(), rd[0]) and np.array_equal(got[1, world - 1, 0].numpy(), rq[0])
            # ... and the first run on real multi-GPU hardware is a PARITY run too: one gathered tile of EVERY rank (its last one, so
            # that the offset inside a rank's block counts) against the float64 oracle of that global tile index -- what arrived over
            # xGMI is the ocean the reference would synthesise for (seed, t), 1e-5 of every channel's maximum
            from oracle import oracle as O
            for r in range(world):
                g = r * per + (per - 1)
                o = O.Oracle(n)
                o.prepare(seed=seed + g)
                _, od, oq = o.compute_waves(0.5, fft=O.FFT_F64)
                for m, want in ((0, od), (1, oq)):
                    have = got[m, r, per - 1].numpy().astype(np.float64)
                    for c in range(4):
                        den = max(float(np.abs(want[..., c]).max()), 1e-30)
                        err = float(np.abs(have[..., c] - want[..., c]).max()) / den
                        if err > 1e-5:
                            print(f"GATHER_PARITY rank {r} tile {g} map {m} channel {c}: {err:.3e}", flush=True)
                            ok = False
    b.comm_destroy(); b.close()
    dist.barrier()
    if rank == 0:
        print("GATHER_OK" if ok else "GATHER_MISMATCH", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
