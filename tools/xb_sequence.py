"""bench.py's context sequence, repeated in one process, with per-kernel serial times and buffer addresses after each step -- the setting in
which the slow serial k_xpass_b showed up (a pipelined context of three chains is used and closed, then a fresh serial context is timed).
Needs a developer build.  usage: xb_sequence.py [rounds] [empty_cache]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import watersurfacerendering_amd as W
from watersurfacerendering_amd import _abi
n = 2048
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
empty = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = _abi.lib()
L.ocean_debug_buffers.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
torch.zeros(1, device="cuda:0")
def timed(b, tag):
    b.time_frames(0.0, 0.05, 100, 50, per_kernel=False)
    ms, k = b.time_frames(0.0, 0.05, 100, 200)
    ptr = (C.c_void_p * 8)()
    L.ocean_debug_buffers(b._h, 0, ptr)
    p = [x or 0 for x in ptr]
    print(f"{tag:14s} frame {ms / 200 * 1e3:6.1f} us  z {k[0]*1e3:6.2f}  xb {k[1]*1e3:6.2f}  disp {k[2]*1e3:6.2f}   z {p[2]:#x} zh {p[3]:#x} hraw {p[4]:#x} maps {p[6]:#x}", flush=True)
for r in range(rounds):
    b = W.OceanBatch(n, 1, 0)
    b.set_pipeline_depth(3)
    b.prepare(0x5EED0000)
    for j in range(1500):
        b.compute_waves_async(0.05 * j)
    b.synchronize()
    b.set_pipeline_depth(1)
    timed(b, f"r{r} bench ctx")
    b.close()
    if empty:
        torch.cuda.empty_cache()
    s = W.OceanBatch(n, 1, 0)
    s.prepare(0x5EED0000)
    timed(s, f"r{r} serial ctx")
    s.close()
