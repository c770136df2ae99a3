#!/usr/bin/env python3
"""Developer probe: how long after a process starts (or after an idle gap) do a serial 2048^2 frame's kernels run at their steady-state
durations?  Windows of 50 serial frames (per-kernel times from dispatch-attached events), back to back from the first frame on; then the same
again after sleeping.     python3 tools/clock_ramp.py [N] [windows] [idle_seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: E402,F401
import watersurfacerendering_amd as W  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
windows = int(sys.argv[2]) if len(sys.argv) > 2 else 40
idle = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
b = W.OceanBatch(n, 1, 0)
b.prepare(0x5EED0000)
for phase in ("from process start", f"after {idle:.0f} s idle", f"after {idle:.0f} s idle again"):
    t0 = time.perf_counter()
    line = []
    for w in range(windows):
        ms, k = b.time_frames(0.0, 0.05, 0, 50, per_kernel=True)
        line.append((time.perf_counter() - t0, k[0] * 1e3, k[1] * 1e3, k[2] * 1e3))
    print(phase)
    for t, z, xb, xd in line:
        print(f"   t = {t * 1e3:7.1f} ms   z {z:6.2f}  xb {xb:6.2f}  xd {xd:6.2f}")
    time.sleep(idle)
b.close()
