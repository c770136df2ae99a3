import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import watersurfacerendering_amd as W
n = int(sys.argv[1]); depths = [int(x) for x in sys.argv[2].split(",")]
out = []
for depth in depths:
    b = W.OceanBatch(n, 1, 0); b.prepare(1); b.set_pipeline_depth(depth)
    frames = 300 if n <= 2048 else 60
    ms, _ = b.time_frames(0.0, 0.05, 20, frames, per_kernel=False)
    out.append(f"d{depth}={ms/frames*1e3:.1f}")
    b.close()
print(f"N={n} " + " ".join(out))
