"""Quick per-size timing of the HIP pipeline (developer tool, not the bench contract)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import watersurfacerendering_amd as W

sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [512, 1024, 2048, 4096]
tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for n in sizes:
    b = W.OceanBatch(n, tiles, 0)
    b.prepare(0x5EED0000)
    frames = 200 if n <= 1024 else (100 if n == 2048 else 30)
    ms, k = b.time_frames(0.0, 0.05, 10, frames)
    per = ms / frames
    gb = 108.0 * n * n * tiles / (per * 1e-3) / 1e9
    print(f"N={n} tiles={tiles}: {per*1e3:.1f} us/frame  {1e3/per:.0f} frames/s  {n*n*tiles/per/1e6:.2f} Gtexel/s  "
          f"alg {gb:.0f} GB/s ({gb/8000*100:.1f}% of 8TB/s)  kernels(us): zpass {k[0]*1e3:.1f} height {k[1]*1e3:.1f} maps {k[2]*1e3:.1f}")
    b.close()
