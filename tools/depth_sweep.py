import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import watersurfacerendering_amd as W
for n in [int(x) for x in sys.argv[1].split(",")]:
    for depth in (1, 2, 3, 4):
        b = W.OceanBatch(n, 1, 0); b.prepare(1); b.set_pipeline_depth(depth)
        frames = 200 if n <= 2048 else 40
        ms, _ = b.time_frames(0.0, 0.05, 10, frames, per_kernel=False)
        print(f"N={n} depth={depth}: {ms/frames*1e3:.1f} us/frame")
        b.close()
