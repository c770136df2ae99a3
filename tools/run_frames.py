"""Runs a few frames of one configuration (profiling target for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: F401  (OCEAN_HIP_LIB -> variant library, developer A/B only)
import watersurfacerendering_amd as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tiles = int(sys.argv[3]) if len(sys.argv) > 3 else 1
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 1          # pipeline depth: 3 = the regime bench.py's headline times
b = W.OceanBatch(n, tiles, 0)
b.set_pipeline_depth(depth)
b.prepare(0x5EED0000)
for j in range(frames):
    b.compute_waves_async(0.05 * j)
b.synchronize()
b.close()
