"""Soak: many pipelined frames, then the last frame must equal a fresh context's (developer tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: F401  (OCEAN_HIP_LIB -> variant library, developer A/B only)
import watersurfacerendering_amd as W
for n, frames, depth in [(2048, 30000, 3), (512, 200000, 4), (1024, 50000, 2)]:
    b = W.OceanBatch(n, 1, 0); b.prepare(9); b.set_pipeline_depth(depth)
    t0 = time.perf_counter()
    for j in range(frames):
        b.compute_waves_async(0.01 * (j % 1000))
    b.synchronize()
    dt = time.perf_counter() - t0
    tl = 0.01 * ((frames - 1) % 1000)
    d, q = b.read_maps(); a = b.heights(0)
    f = W.OceanBatch(n, 1, 0); f.prepare(9); a2 = f.compute_waves(tl); d2, q2 = f.read_maps()
    print(f"N={n} depth={depth}: {frames} frames in {dt:.2f} s ({dt/frames*1e6:.1f} us/frame); last frame identical to a fresh context: "
          f"{np.array_equal(d, d2) and np.array_equal(q, q2) and a[0] == a2[0]}")
    b.close(); f.close()
