"""Runs a few frames of one configuration (profiling target for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import watersurfacerendering_amd as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tiles = int(sys.argv[3]) if len(sys.argv) > 3 else 1
b = W.OceanBatch(n, tiles, 0)
b.prepare(0x5EED0000)
for j in range(frames):
    b.compute_waves_async(0.05 * j)
b.synchronize()
b.close()
