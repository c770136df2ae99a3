"""bench.py's main sequence step by step, serial per-kernel times after every stage (developer tool)"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.set_device(0)
import watersurfacerendering_amd as W
def serial(tag, b):
    ms, k = b.time_frames(0.0, 0.05, 200, 200)
    print(f"{tag:58s} serial {ms/200*1e3:6.1f} us  " + "  ".join(f"{v*1e3:6.2f}" for v in k), flush=True)
use_torch_sync = "--no-torch-sync" not in sys.argv
b = W.OceanBatch(2048, 1, 0); b.set_pipeline_depth(3); b.prepare(0x5EED0000)
def sync():
    b.synchronize()
    if use_torch_sync: torch.cuda.synchronize()
for j in range(500): b.compute_waves_async(0.05 * j)
sync()
for j in range(1000): b.compute_waves_async(0.05 * j)
sync(); sync()
t0 = time.perf_counter()
for j in range(2000): b.compute_waves_async(0.05 * (1000 + j))
sync(); sync()
print(f"timed region {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us/step", flush=True)
b.set_pipeline_depth(1); serial("right after the timed region (no pipelined kernel pass)", b); b.set_pipeline_depth(3)
_, kp = b.time_frames(0.0, 0.05, 200, 200)
b.set_pipeline_depth(1)
serial("after the pipelined per-kernel pass", b)
time.sleep(1.0)
serial("after sleeping 1 s", b)
b.close(); torch.cuda.empty_cache()
c = W.OceanBatch(2048, 1, 0); c.prepare(0x5EED0000)
serial("fresh context", c)
c.close()
