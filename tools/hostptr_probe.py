#!/usr/bin/env python3
"""Round 6 debugging aid: does the runtime hand out a device address for PAGEABLE host memory that it pinned on the fly for an earlier copy?
(ocean_compute_waves_read's direct-store path must only ever take memory that is page-locked for as long as the caller says.)
Runs frames + read-outs into pageable arrays of several sizes, frees them, allocates again and asks hipHostGetDevicePointer each time."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: E402,F401
import watersurfacerendering_amd as W  # noqa: E402

hip = C.CDLL("libamdhip64.so")
hip.hipHostGetDevicePointer.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint]


def dev_ptr(a):
    p = C.c_void_p()
    e = hip.hipHostGetDevicePointer(C.byref(p), a.ctypes.data_as(C.c_void_p), 0)
    hip.hipGetLastError()
    return e, p.value


tiles = 2
for n in (64, 256, 512, 1024):
    b = W.OceanBatch(n, tiles, 0)
    b.prepare(1)
    found = 0
    for it in range(40):
        d = np.empty((tiles, n, n, 4), np.float32); q = np.empty_like(d)
        e0, p0 = dev_ptr(d)
        if e0 == 0:
            found += 1
            print(f"  n={n} it={it}: hipHostGetDevicePointer SUCCEEDS on a fresh pageable array at {d.ctypes.data:#x} -> {p0:#x}", flush=True)
        amp, _, _ = b.compute_waves_read(0.1 * it, d, q)
        d2, q2 = b.read_maps()
        if not (np.array_equal(d, d2) and np.array_equal(q, q2)):
            print(f"  n={n} it={it}: WRONG maps", flush=True)
        if it % 3 == 0:
            dd = np.empty_like(d); qq = np.empty_like(d)
            b.read_maps_async(dd, qq); b.synchronize()
            del dd, qq
        del d, q
    print(f"n={n}: {found} of 40 fresh pageable arrays had a device address", flush=True)
    b.close()
print("HOSTPTR_PROBE_DONE")
