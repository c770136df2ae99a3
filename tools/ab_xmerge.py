#!/usr/bin/env python3
"""Developer A/B: the merged x pass (k_xpass_b with its DISP workgroups, two launches per frame) against the three-launch frame, interleaved
on one box.  Sizes up to 512 switch through the ABI (ocean_set_merged_xpass); larger ones need a developer build (OCEAN_HIP_LIB=...libocean_hip_dev.so),
whose launcher reads OCEAN_XMERGE per frame.
    python3 tools/ab_xmerge.py N [tiles] [depth] [frames] [repeats]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: E402,F401
import watersurfacerendering_amd as W  # noqa: E402

n = int(sys.argv[1])
tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 1
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 1
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = bool(os.environ.get("OCEAN_HIP_LIB"))
ctx, maps = {}, {}
forms = (0, 1, 2) if dev else (0, 2)     # 0 three launches, 1 merged x pass (two launches; developer build: OCEAN_ONE_LAUNCH=0), 2 what the library picks with the switch on
for merged in forms:
    b = W.OceanBatch(n, tiles, 0)
    b.set_pipeline_depth(depth)
    b.set_merged_xpass(bool(merged))
    b.prepare(0x5EED0000)
    ctx[merged] = b


def select(merged):
    if dev:
        os.environ["OCEAN_XMERGE"] = "1" if merged else "0"
        os.environ["OCEAN_ONE_LAUNCH"] = "1" if merged == 2 else "0"


for rep in range(reps):
    for merged in forms:
        b = ctx[merged]
        select(merged)
        ms, kern = b.time_frames(0.0, 0.05, frames // 3, frames, per_kernel=True)
        fl = b.last_launch()[1]["flags"]
        flag = "one launch" if fl & 4096 else ("merged x" if fl & 1024 else "three")
        sync_us = None
        if depth == 1:
            for j in range(50):
                b.compute_waves(0.05 * j)
            ts = np.empty(400)
            for j in range(400):
                t0 = time.perf_counter()
                b.compute_waves(0.05 * j)
                ts[j] = time.perf_counter() - t0
            sync_us = float(np.median(ts) * 1e6)
        print(f"N={n} tiles={tiles} depth={depth} form={flag:10} {ms / frames * 1e3:8.2f} us/frame   kernels {' / '.join(f'{k * 1e3:.2f}' for k in kern)} us"
              + (f"   synchronous call {sync_us:.1f} us" if sync_us else ""), flush=True)
for merged in forms:
    b = ctx[merged]
    select(merged)
    b.set_pipeline_depth(1)
    b.compute_waves(1.25)
    maps[merged] = b.read_maps()
    b.close()
print("bit-identical:", all(np.array_equal(x, y) for f in forms[1:] for x, y in zip(maps[0], maps[f])))
