#!/usr/bin/env python3
"""Developer probe for the slow-context mode of the serial z pass: bench.py's sequence in a loop -- a context runs pipelined frames and is
closed, a NEW context measures serial per-kernel times -- with the device addresses of the new context's buffers (developer build:
ocean_debug_buffers) beside the times: does a slow context sit on differently aligned / placed memory?
    OCEAN_HIP_LIB=...libocean_hip_dev.so python3 tools/ctx_churn.py [rounds]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: E402,F401
import watersurfacerendering_amd as W  # noqa: E402
from watersurfacerendering_amd import _abi  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
L = _abi.lib()
dbg = getattr(L, "ocean_debug_buffers", None)
for r in range(rounds):
    a = W.OceanBatch(2048, 1, 0)
    a.set_pipeline_depth(3)
    a.prepare(0x5EED0000)
    a.time_frames(0.0, 0.05, 50, 150 + 40 * r, per_kernel=False)
    a.close()
    if r % 2:
        import torch
        torch.cuda.empty_cache()
    b = W.OceanBatch(2048, 1, 0)
    b.prepare(0x5EED0000)
    ms, k = b.time_frames(0.0, 0.05, 300, 200, per_kernel=True)
    addr = ""
    if dbg:
        out = (C.c_void_p * 8)()
        dbg.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        if dbg(b._h, 0, out) == 0:
            names = ["h0", "omega_q", "z", "zh", "hraw", "minmax", "disp", "nrm"]
            addr = "  ".join(f"{nm}={(out[i] or 0):#x}" for i, nm in enumerate(names) if nm in ("h0", "z", "zh", "disp"))
    print(f"round {r:2d}: frame {ms / 200 * 1e3:6.2f} us   z {k[0] * 1e3:6.2f}  xb {k[1] * 1e3:6.2f}  xd {k[2] * 1e3:6.2f}   {addr}", flush=True)
    b.close()
