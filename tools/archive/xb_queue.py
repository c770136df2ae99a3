"""Probe for the slow serial k_xpass_b: does the hardware queue a context's streams land on matter?  k extra HIP streams are created (and kept)
ahead of the context, which shifts the round-robin assignment of its streams to the process's hardware queues; then the usual per-kernel timing.
usage: xb_queue.py [contexts per k]"""
import ctypes as C, os, sys, time
T0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: F401  (OCEAN_HIP_LIB -> variant library, developer A/B only)
import watersurfacerendering_amd as W
hip = C.CDLL("libamdhip64.so")
from watersurfacerendering_amd import _abi
L = _abi.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
keep = []
for k in [0] + [1] * int(os.environ.get('XBQ_MAX', '7')):
    for _ in range(k):
        s = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0; keep.append(s)
    for r in range(reps):
        b = W.OceanBatch(2048, 1, 0); b.prepare(0x5EED0000)
        b.time_frames(0.0, 0.05, 200, 50, per_kernel=False)
        ms, kk = b.time_frames(0.0, 0.05, 100, 200)
        addr = ""
        if hasattr(L, "ocean_debug_buffers"):          # developer build: device addresses beside the timings
            ptr = (C.c_void_p * 8)()
            L.ocean_debug_buffers.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
            L.ocean_debug_buffers(b._h, 0, ptr)
            addr = "  " + " ".join(f"{nm} {(x or 0):#x}" for nm, x in zip(["h0", "omega_q", "z", "zh", "hraw", "minmax", "disp", "nrm"], ptr) if nm != "minmax")
        print(f"{time.perf_counter() - T0:6.2f} s extra streams {len(keep)}: serial {ms/200*1e3:6.1f} us/frame  " + "  ".join(f"{nm} {v*1e3:6.2f}" for nm, v in zip(b.kernel_names(), kk)) + addr, flush=True)
        b.close()
