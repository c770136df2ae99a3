#!/usr/bin/env python3
"""Round 6: what ocean_prepare's placement search buys.  Contexts created one after the other in ONE process (all alive), alternately without
and with the search; per context the serial z pass / frame time (dispatch-attached events, 300 frames behind 60 ms of load) and the search's
own report.  Without it a context's speed is whatever its allocation drew; with it every context should sit at the fast end.
    python3 tools/placement_probe.py [N] [contexts] [tiles] [candidates when on: 0 = the library's rule]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: E402,F401
import watersurfacerendering_amd as W  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
tiles = int(sys.argv[3]) if len(sys.argv) > 3 else 1
forced = int(sys.argv[4]) if len(sys.argv) > 4 else 0
alive = []
rows = {0: [], 1: []}
for k in range(count):
    on = k % 2
    b = W.OceanBatch(n, tiles, 0)
    b.set_placement_search(forced if on else 1)
    t0 = time.perf_counter()
    b.prepare(0x5EED0000 + k)
    prep_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.06:
        for j in range(50):
            b.compute_waves_async(0.05 * j)
        b.synchronize()
    ms, kern = b.time_frames(0.0, 0.05, 50, 300, per_kernel=True)
    tried, chosen, worst = b.placement_report()
    rows[on].append((kern[0] * 1e3, ms / 300 * 1e3))
    print(f"context {k:2d} search {'on ' if on else 'off'}: prepare {prep_ms:6.1f} ms  z pass {kern[0] * 1e3:6.2f} us  serial frame {ms / 300 * 1e3:6.1f} us   "
          f"candidates {tried}  chosen {chosen:6.1f} us  slowest {worst:6.1f} us", flush=True)
    alive.append(b)
for on in (0, 1):
    z = sorted(r[0] for r in rows[on])
    print(f"search {'on ' if on else 'off'}: z pass min {z[0]:.2f}  median {z[len(z) // 2]:.2f}  max {z[-1]:.2f} us over {len(z)} contexts")
for b in alive:
    b.close()
