#!/usr/bin/env python3
"""Round 6: the launch heuristics as RULES (OceanTuning, csrc/ocean_ctx.h) against the choices they replace.
The staggered start is now given to every launch whose whole grid is resident at once and that moves >= 64 MB, with a spread of
0.27 x bytes / 5.5 TB/s; ocean_set_start_ramp(ctx, 0) switches it off in the same library, so each configuration is timed both ways,
interleaved, three repeats: serial frames (per kernel) and pipelined ones.  Where the rule gives no ramp the two columns must agree
(noise); where it gives one -- 2048^2 x 1 as before, NEW: 3-5 tiles of 1024^2 -- "on" must not lose.
    python3 tools/ab_tuning.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: E402,F401
import watersurfacerendering_amd as W  # noqa: E402
A = W._abi

CONFIGS = [(1024, 1, 3), (1024, 3, 2), (1024, 4, 2), (1024, 5, 2), (2048, 1, 3), (2048, 2, 2), (4096, 1, 2)]
for n, tiles, depth in CONFIGS:
    rows = {True: [], False: []}
    flags = {}
    for rep in range(3):
        for on in (True, False):
            b = W.OceanBatch(n, tiles, 0)
            b.set_start_ramp(on)
            b.prepare(0x5EED0000)
            frames = 400 if n <= 2048 else 120
            b.time_frames(0.0, 0.05, 200, 50, per_kernel=False)
            ms, k = b.time_frames(0.0, 0.05, 100, frames)
            flags[on] = [bool(li["flags"] & A.OCEAN_LAUNCH_STAGGERED_START) for li in b.last_launch()]
            b.set_pipeline_depth(depth)
            msp, _ = b.time_frames(0.0, 0.05, 300, 2 * frames, per_kernel=False)
            rows[on].append((ms / frames * 1e3, [v * 1e3 for v in k], msp / (2 * frames) * 1e3))
            b.close()
    for on in (True, False):
        r = rows[on]
        print(f"N={n}x{tiles} ramp {'on ' if on else 'off'} staggered launches {flags[on]}  serial " + " / ".join(f"{x[0]:6.1f}" for x in r) +
              "  z " + " / ".join(f"{x[1][0]:5.2f}" for x in r) + "  xb " + " / ".join(f"{x[1][1]:5.2f}" for x in r) +
              "  xd " + " / ".join(f"{x[1][2]:5.2f}" for x in r) + f"  depth {depth}: " + " / ".join(f"{x[2]:6.1f}" for x in r), flush=True)
