#!/usr/bin/env python3
"""Soak of the merged x pass's in-launch hand-off (HEIGHT workgroups -> DISP workgroups: write-through stores, one counted arrival per
workgroup, a polled wait): random tile sizes <= 512, pipeline depths 2..8, modes, precisions, bursts of random length -- the last frame of
every burst (amplitude, heights, both maps) against a context with the three-launch frame fed the same times.  A stale read would show here.
usage: soak_merged.py [seconds] [seed]"""
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: E402,F401
import watersurfacerendering_amd as W  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
frames = bursts = configs = 0
while time.time() < t_end:
    n = rng.choice([16, 32, 64, 128, 256, 512, 512])
    depth = rng.choice([2, 3, 4, 8])
    mode = rng.choice([0, 0, 1, 2])
    bits = rng.choice([32, 32, 16])
    ctx = []
    for merged in (True, False):
        b = W.OceanBatch(n, 1, 0)
        b.set_mode(mode); b.set_intermediate_precision(bits); b.set_pipeline_depth(depth); b.set_merged_xpass(merged); b.set_frame_tracking(rng.random() < 0.5)
        b.prepare(1000 + configs)
        ctx.append(b)
    configs += 1
    for _ in range(rng.randint(20, 60)):
        k = rng.randint(1, 4 * depth)
        times = [rng.uniform(0.0, 2000.0) for _ in range(k)]
        out = []
        for b in ctx:
            for t in times:
                b.compute_waves_async(t)
            amp = b.wait_frame()
            d, q = b.read_maps()
            out.append((amp.copy(), b.heights(0), d.copy(), q.copy()))
        flags = ctx[0].last_launch()[1]["flags"]
        assert flags & 1024, "the merged form did not run"
        assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1], (n, depth, mode, bits, bursts)
        assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][3], out[1][3]), (n, depth, mode, bits, bursts)
        frames += k
        bursts += 1
    for b in ctx:
        b.close()
print(f"soak_merged: {configs} configurations, {bursts} bursts, {frames} frames per context, all identical to the three-launch frame")
