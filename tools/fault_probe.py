#!/usr/bin/env python3
"""Developer check of the in-launch waits' failure path: on a variant build whose DISP workgroups wait for one HEIGHT arrival too many
(make -C watersurfacerendering_amd/csrc variant NAME=fault DEFS=-DOCEAN_FAULT_INJECT), a merged frame's wait must give up after 20 ms -- no
hang -- and the host must report an error from the next wait / synchronisation / read-out instead of handing out the frame.
    OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_fault.so python3 tools/fault_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: E402,F401
import watersurfacerendering_amd as W  # noqa: E402

b = W.OceanBatch(512, 1, 0)
b.set_pipeline_depth(2)
b.prepare(3)
t0 = time.perf_counter()
b.compute_waves_async(0.5)                      # pipelined 512^2 frame: merged x pass -> the injected wait
try:
    b.synchronize()
    print("NO ERROR reported: FAIL")
    sys.exit(1)
except W.OceanError as e:
    print(f"error reported after {(time.perf_counter() - t0) * 1e3:.1f} ms: {e}")
try:
    b.read_maps()
    print("read-out handed out the frame: FAIL")
    sys.exit(1)
except W.OceanError as e:
    print("read-out refused:", e.code)
b.prepare(3)                                    # the fault is sticky until the next Prepare, which drains the context and clears it
b.set_pipeline_depth(1)                         # a serial 512^2 frame keeps three launches: no in-launch wait
amp = b.compute_waves(0.5)
print("serial frame after re-Prepare fine: A =", float(amp[0]))
b.close()
print("FAULT_PATH_OK")
