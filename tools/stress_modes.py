"""Stress: every combination of mode x intermediate precision x pipeline depth x batch on one context sequence must keep
producing the maps a fresh single-purpose context produces (developer tool)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: F401  (OCEAN_HIP_LIB -> variant library, developer A/B only)
import watersurfacerendering_amd as W
bad = 0
for n, tiles in [(64, 5), (512, 2), (2048, 1)]:
    b = W.OceanBatch(n, tiles, 0)
    for bits, mode, depth in itertools.product((32, 16, 32), (0, 3, 1, 2, 0), (1, 3)):
        b.set_intermediate_precision(bits); b.set_mode(mode); b.set_pipeline_depth(depth)
        b.prepare(123)
        for j in range(4): b.compute_waves_async(0.2 * j)
        b.compute_waves_async(1.7); b.synchronize()
        d, q = b.read_maps(tiles - 1, 1)
        f = W.OceanBatch(n, 1, 0); f.set_intermediate_precision(bits); f.set_mode(mode); f.prepare(123 + tiles - 1)
        f.compute_waves(1.7); d2, q2 = f.read_maps(); f.close()
        ok = np.array_equal(d, d2) and np.array_equal(q, q2)
        bad += not ok
        if not ok: print("MISMATCH", n, tiles, bits, mode, depth)
    b.close()
print("stress_modes:", "all identical" if bad == 0 else f"{bad} mismatches")
