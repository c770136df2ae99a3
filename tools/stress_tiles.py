import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: F401  (OCEAN_HIP_LIB -> variant library, developer A/B only)
import watersurfacerendering_amd as W
for n, tiles in [(16, 4096), (64, 2048), (256, 256), (1024, 16)]:
    b = W.OceanBatch(n, tiles, 0); b.prepare(77)
    b.set_pipeline_depth(2)
    for j in range(4): b.compute_waves_async(0.3 * j)
    b.synchronize()
    amp = b.compute_waves(1.25)
    d, q = b.read_maps(tiles - 2, 2)
    s = W.OceanBatch(n, 1, 0); s.prepare(77 + tiles - 1); a1 = s.compute_waves(1.25); d1, q1 = s.read_maps()
    ok = np.array_equal(d[1], d1[0]) and np.array_equal(q[1], q1[0]) and amp[-1] == a1[0]
    print(n, tiles, "last tile identical to a single-tile run:", ok, "amp range", float(amp.min()), float(amp.max()))
    b.close(); s.close()
