import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import watersurfacerendering_amd as W
n, tiles = int(sys.argv[1]), int(sys.argv[2])
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 2
pre = int(sys.argv[4]) if len(sys.argv) > 4 else 4
b = W.OceanBatch(n, tiles, 0); b.prepare(77)
b.set_pipeline_depth(depth)
for j in range(pre): b.compute_waves_async(0.3 * j)
b.synchronize()
amp = b.compute_waves(1.25)
d, q = b.read_maps()
bad_tiles = []
for i in list(range(0, tiles, max(1, tiles // 16))) + [tiles - 1]:
    s = W.OceanBatch(n, 1, 0); s.prepare(77 + i); a1 = s.compute_waves(1.25); d1, q1 = s.read_maps(); s.close()
    if not (np.array_equal(d[i], d1[0]) and np.array_equal(q[i], q1[0]) and amp[i] == a1[0]):
        dd = np.abs(d[i] - d1[0]); dq = np.abs(q[i] - q1[0])
        rows = np.unique(np.nonzero((dd.max(-1) > 0) | (dq.max(-1) > 0))[0])
        bad_tiles.append((i, float(dd.max()), float(dq.max()), int((dd > 0).sum()), int((dq > 0).sum()), float(np.abs(d1).max()), amp[i], a1[0], rows[:8], len(rows)))
print(n, tiles, depth, pre, "mismatching tiles:", bad_tiles)
b.close()
