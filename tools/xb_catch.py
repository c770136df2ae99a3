"""Tries to catch a slow-k_xpass_b context in the act and, when it has one, asks it questions: bench.py's exact sequence up to
its serial pass; if the fresh serial context's k_xpass_b is more than 10 % slower than the bench context's, (1) its maps are re-bound to freshly
allocated memory (ocean_bind_output) and it is timed again -- normal then: the normal map's backing memory was the cause; (2) un-bound and timed
again; (3) another context is created beside it and timed.  Developer build for the addresses.  usage: xb_catch.py [tag]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import watersurfacerendering_amd as W
from watersurfacerendering_amd import _abi
n, SEED, DT = 2048, 0x5EED0000, 0.05
tag = sys.argv[1] if len(sys.argv) > 1 else ""
L = _abi.lib()
have_dbg = hasattr(L, "ocean_debug_buffers")
if have_dbg:
    L.ocean_debug_buffers.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
torch.cuda.set_device(0)
def addrs(b):
    if not have_dbg: return ""
    ptr = (C.c_void_p * 8)(); L.ocean_debug_buffers(b._h, 0, ptr); p = [x or 0 for x in ptr]
    return f"z {p[2]:#x} zh {p[3]:#x} hraw {p[4]:#x} maps {p[6]:#x}"
def kern(b, warm=300, frames=200):
    ms, k = b.time_frames(0.0, DT, warm, frames, per_kernel=True)
    return [x * 1e3 for x in k]
b = W.OceanBatch(n, 1, 0); b.set_pipeline_depth(3); b.prepare(SEED)
for j in range(1500): b.compute_waves_async(DT * j)
b.synchronize(); torch.cuda.synchronize()
for j in range(2000): b.compute_waves_async(DT * j)
b.synchronize(); torch.cuda.synchronize()
b.time_frames(0.0, DT, 200, 200, per_kernel=True)
b.set_pipeline_depth(1)
kb = kern(b, 200, 200)
b.close(); torch.cuda.empty_cache()
s = W.OceanBatch(n, 1, 0); s.prepare(SEED)
ks = kern(s)
slow = ks[1] > 1.10 * kb[1] or kb[1] > 23.5
print(f"{tag} bench ctx z {kb[0]:.2f} xb {kb[1]:.2f} disp {kb[2]:.2f} | serial ctx z {ks[0]:.2f} xb {ks[1]:.2f} disp {ks[2]:.2f} | {'SLOW' if slow else 'normal'}  {addrs(s)}", flush=True)
if slow:
    maps = torch.zeros((2, n, n, 4), dtype=torch.float32, device="cuda:0")
    s.bind_output(maps[0].data_ptr(), maps[1].data_ptr())
    k1 = kern(s, 100, 200)
    print(f"{tag}   maps re-bound to fresh memory ({maps.data_ptr():#x}): z {k1[0]:.2f} xb {k1[1]:.2f} disp {k1[2]:.2f}", flush=True)
    s.bind_output(None, None)
    k2 = kern(s, 100, 200)
    print(f"{tag}   un-bound again:                        z {k2[0]:.2f} xb {k2[1]:.2f} disp {k2[2]:.2f}", flush=True)
    s2 = W.OceanBatch(n, 1, 0); s2.prepare(SEED)
    k3 = kern(s2)
    print(f"{tag}   a second context beside it:            z {k3[0]:.2f} xb {k3[1]:.2f} disp {k3[2]:.2f}  {addrs(s2)}", flush=True)
    k4 = kern(s, 100, 200)
    print(f"{tag}   the slow context once more:            z {k4[0]:.2f} xb {k4[1]:.2f} disp {k4[2]:.2f}", flush=True)
    s2.close()
s.close()
