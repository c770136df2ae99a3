import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import watersurfacerendering_amd as W
torch.cuda.set_device(0)
for depth in (1, 2):
    b = W.OceanBatch(2048, 1, 0); b.set_pipeline_depth(depth); b.prepare(1)
    for j in range(10): b.compute_waves_async(0.05 * j)
    b.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(200): b.compute_waves_async(0.05 * j)
    t1 = time.perf_counter()
    b.synchronize(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    ms, _ = b.time_frames(0.0, 0.05, 10, 200, per_kernel=False)
    print(f"depth {depth}: python loop {1e6*(t2-t0)/200:.1f} us/frame (enqueue {1e6*(t1-t0)/200:.1f}), C loop {ms/200*1e3:.1f}")
    b.close()
