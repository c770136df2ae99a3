"""Host enqueue cost vs GPU time of asynchronous frames (developer tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import watersurfacerendering_amd as W
for n in [int(x) for x in sys.argv[1].split(",")]:
    for depth in (1, 4):
        b = W.OceanBatch(n, 1, 0); b.prepare(1); b.set_pipeline_depth(depth)
        for j in range(50): b.compute_waves_async(0.05 * j)
        b.synchronize()
        frames = 2000
        t0 = time.perf_counter()
        for j in range(frames): b.compute_waves_async(0.05 * j)
        t1 = time.perf_counter()
        b.synchronize()
        t2 = time.perf_counter()
        ms, _ = b.time_frames(0.0, 0.05, 10, frames, per_kernel=False)
        print(f"N={n} depth={depth}: python enqueue {1e6*(t1-t0)/frames:.1f} us/frame, enqueue+drain {1e6*(t2-t0)/frames:.1f} us/frame, C-loop time_frames {ms/frames*1e3:.1f} us/frame")
        b.close()
