#!/usr/bin/env python3
"""Developer probe: per-kernel times of serial frames over MANY fresh contexts in one process (a context = its own buffers and streams), with the
write-through z pass on and off (developer build: OCEAN_Z_WT read per frame) -- is a kernel's time a property of the code or of the context?
    OCEAN_HIP_LIB=...libocean_hip_dev.so python3 tools/ctx_spread.py [N] [contexts] [frames]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: E402,F401
import watersurfacerendering_amd as W  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
contexts = int(sys.argv[2]) if len(sys.argv) > 2 else 12
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 200
keep = []
for i in range(contexts):
    b = W.OceanBatch(n, 1, 0)
    b.prepare(0x5EED0000 + i)
    out = []
    for wt in (1, 0, 1, 0):
        os.environ["OCEAN_Z_WT"] = str(wt)
        ms, k = b.time_frames(0.0, 0.05, 100, frames, per_kernel=True)
        out.append(f"wt={wt}: z {k[0] * 1e3:6.2f} xb {k[1] * 1e3:6.2f} xd {k[2] * 1e3:6.2f}")
    print(f"context {i:2d}  " + "   ".join(out), flush=True)
    if i % 3 == 2:
        keep.append(b)          # some contexts stay alive: the next ones get other memory
    else:
        b.close()
for b in keep:
    b.close()
