#!/usr/bin/env python3
"""Developer A/B: standard against split frame order (profiles/r05_4096_experiments.txt), interleaved on one box.  Needs a developer
build of the library (make -C watersurfacerendering_amd/csrc variant NAME=dev DEFS=-DOCEAN_DEVELOPER; tools/devlib.py points the loader at it):
the order is chosen per frame from the environment variable OCEAN_FRAME_ORDER, which this script flips between the passes.
    python3 tools/ab_order.py N tiles depth [frames] [repeats] [inter_bits]
Prints us per frame (ocean_time_frames, whole region) and the per-kernel sums of a serial pass, and checks that the two orders deliver the same bits."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: E402,F401  (OCEAN_HIP_LIB -> the developer build)
import watersurfacerendering_amd as W  # noqa: E402

n, tiles, depth = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 300
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
bits = int(sys.argv[6]) if len(sys.argv) > 6 else 32
ctx, maps = {}, {}
for order in (1, 2):
    b = W.OceanBatch(n, tiles, 0)
    b.set_intermediate_precision(bits)
    b.set_pipeline_depth(depth)
    b.prepare(0x5EED0000)
    ctx[order] = b
for rep in range(reps):
    for order in (1, 2):
        b = ctx[order]
        os.environ["OCEAN_FRAME_ORDER"] = str(order)
        ms, kern = b.time_frames(0.0, 0.05, frames // 3, frames, per_kernel=True)
        flags = b.last_launch()[0]["flags"]
        print(f"N={n} tiles={tiles} depth={depth} bits={bits} order={'standard' if order == 1 else 'split   '} split_flag={bool(flags & 512)} "
              f"{ms / frames * 1e3:8.2f} us/frame   kernels {' / '.join(f'{k * 1e3:.2f}' for k in kern)} us", flush=True)
for order in (1, 2):
    b = ctx[order]
    os.environ["OCEAN_FRAME_ORDER"] = str(order)
    b.set_pipeline_depth(1)
    b.compute_waves(1.25)
    maps[order] = b.read_maps()
    b.close()
print("bit-identical:", all(np.array_equal(x, y) for x, y in zip(maps[1], maps[2])))
