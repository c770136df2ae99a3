"""Placement probe for the slow serial k_xpass_b: contexts created one after another in ONE process, each timed
(serial 2048^2 frames, per-kernel execution times) with the device addresses of its buffers beside it.  Needs a developer build
(OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_dev.so).  usage: xb_placement.py [contexts] [perturb]
perturb = 1: a torch allocation of a varying size is made (and kept) between contexts, so that the next context's buffers land elsewhere."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import watersurfacerendering_amd as W
from watersurfacerendering_amd import _abi
n = 2048
count = int(sys.argv[1]) if len(sys.argv) > 1 else 10
perturb = int(sys.argv[2]) if len(sys.argv) > 2 else 0
L = _abi.lib()
L.ocean_debug_buffers.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
keep = []
names = ["h0", "omega_q", "z", "zh", "hraw", "minmax", "disp", "nrm"]
for i in range(count):
    if perturb:
        keep.append(torch.empty(((i * 7919) % 61 + 1) * 1024 * 1024 + (i % 5) * 4096, dtype=torch.uint8, device="cuda:0"))
    b = W.OceanBatch(n, 1, 0)
    b.prepare(0x5EED0000)
    b.time_frames(0.0, 0.05, 100, 50, per_kernel=False)
    ms, k = b.time_frames(0.0, 0.05, 50, 200)
    ptr = (C.c_void_p * 8)()
    L.ocean_debug_buffers(b._h, 0, ptr)
    p = [x or 0 for x in ptr]
    print(f"ctx {i:2d}  frame {ms / 200 * 1e3:6.1f} us  z {k[0]*1e3:6.2f}  xb {k[1]*1e3:6.2f}  disp {k[2]*1e3:6.2f}   " +
          "  ".join(f"{nm} {v:#x}" for nm, v in zip(names, p) if nm in ("z", "zh", "hraw", "disp", "nrm")), flush=True)
    if perturb != 2:
        b.close()
    else:
        keep.append(b)
