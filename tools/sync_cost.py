"""Latency of the synchronous ocean_compute_waves (the reference's call shape) per size (developer tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: F401  (OCEAN_HIP_LIB -> variant library, developer A/B only)
import watersurfacerendering_amd as W
for n in [int(x) for x in sys.argv[1].split(",")]:
    b = W.OceanBatch(n, 1, 0); b.prepare(1)
    for j in range(50): b.compute_waves(0.05 * j)
    frames = 1000 if n <= 1024 else 300
    ts = np.empty(frames)
    for j in range(frames):
        t0 = time.perf_counter(); b.compute_waves(0.05 * j); ts[j] = time.perf_counter() - t0
    ms, _ = b.time_frames(0.0, 0.05, 10, frames, per_kernel=False)
    p = np.percentile(ts * 1e6, [5, 50, 95])
    print(f"N={n}: synchronous ComputeWaves mean {ts.mean()*1e6:.1f} us/call, p5/p50/p95 {p[0]:.1f}/{p[1]:.1f}/{p[2]:.1f} (async frames back to back: {ms/frames*1e3:.1f} us/frame)")
    b.close()
