#!/usr/bin/env python3
"""Developer check of the in-launch waits' failure path (round 6: a give-up is RECOVERED, not fatal).  On a variant build whose
DISP workgroups wait for one HEIGHT arrival too many (make -C watersurfacerendering_amd/csrc variant NAME=fault DEFS=-DOCEAN_FAULT_INJECT)
every merged frame's wait gives up after 20 ms -- no hang -- and leaves wrong maps behind.  The host must then, in the SAME call that
notices it: drain, keep the three-launch frame for the context, run the affected frames again and hand out the right frame -- compared here
with a context that never used the merged form.  Cases: synchronize + read-out, the synchronous call at 128^2 (serial merged frame),
several chains in flight, ocean_compute_waves_read, and a stream-ordered consumer behind a faulted frame (reported once, context usable).
    OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_fault.so python3 tools/fault_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: E402,F401
import watersurfacerendering_amd as W  # noqa: E402


def reference(n, seed, t):
    r = W.OceanBatch(n, 1, 0)
    r.set_merged_xpass(False)
    r.prepare(seed)
    a = r.compute_waves(t)
    d, q = r.read_maps()
    r.close()
    return a, d, q


def same(x, y):
    return all(np.array_equal(u, v) for u, v in zip(x, y))


ok = True
# 1. pipelined 512^2 frames (merged x pass -> the injected wait), then synchronize + read-out
b = W.OceanBatch(512, 1, 0)
b.set_pipeline_depth(2)
b.prepare(3)
t0 = time.perf_counter()
b.compute_waves_async(0.5)
b.synchronize()
ms = (time.perf_counter() - t0) * 1e3
got = (b.wait_frame(), *b.read_maps())
want = reference(512, 3, 0.5)
print(f"1. synchronize returned after {ms:.1f} ms, recoveries {b.fault_recoveries}, frame right: {same(got, want)}, merged form now off: "
      f"{not any(li['flags'] & W._abi.OCEAN_LAUNCH_MERGED_X for li in b.last_launch())}")
ok &= same(got, want) and b.fault_recoveries == 1 and ms > 15.0
# ... and the context simply goes on (three launches per frame from now on: no further recovery)
for j in range(6):
    b.compute_waves_async(0.1 * j)
b.synchronize()
got = (b.wait_frame(), *b.read_maps())
print(f"   six more pipelined frames: recoveries {b.fault_recoveries}, last frame right: {same(got, reference(512, 3, 0.5))}")
ok &= same(got, reference(512, 3, 0.5)) and b.fault_recoveries == 1
b.close()

# 2. the synchronous call on a serial merged frame (128^2): the call itself recovers and returns the right amplitude
b = W.OceanBatch(128, 1, 0)
b.prepare(4)
t0 = time.perf_counter()
a = b.compute_waves(1.5)
ms = (time.perf_counter() - t0) * 1e3
got = (a, *b.read_maps())
print(f"2. ocean_compute_waves (128^2 serial, merged) took {ms:.1f} ms, recoveries {b.fault_recoveries}, frame right: {same(got, reference(128, 4, 1.5))}")
ok &= same(got, reference(128, 4, 1.5)) and b.fault_recoveries == 1
b.close()

# 3. three chains in flight, every one of them faulted: all three last frames are run again; the most recent one is what the read-out sees
b = W.OceanBatch(256, 1, 0)
b.set_pipeline_depth(3)
b.prepare(5)
for j in range(3):
    b.compute_waves_async(1.0 + j)
got = (b.wait_frame(), *b.read_maps())
print(f"3. three chains, wait_frame + read-out: recoveries {b.fault_recoveries}, frame right: {same(got, reference(256, 5, 3.0))}")
ok &= same(got, reference(256, 5, 3.0)) and b.fault_recoveries == 1
b.close()

# 4. ocean_compute_waves_read on a merged serial frame: the copies taken from the wrong frame are repeated
b = W.OceanBatch(64, 1, 0)
b.prepare(6)
got = b.compute_waves_read(2.0)
print(f"4. ocean_compute_waves_read (64^2): recoveries {b.fault_recoveries}, frame right: {same(got, reference(64, 6, 2.0))}")
ok &= same(got, reference(64, 6, 2.0)) and b.fault_recoveries == 1
b.close()

# 5. a stream-ordered consumer behind a faulted frame has consumed garbage: reported ONCE by the recovering call, the context stays usable
b = W.OceanBatch(256, 1, 0)
b.prepare(7)
b.compute_waves(0.25); b.build_mips(0)          # (serial 256^2: three launches, no in-launch wait; allocates the mip buffers)
b.set_pipeline_depth(2)
b.compute_waves_async(0.5)                      # pipelined: merged x pass -> the injected wait gives up 20 ms from now
W._abi.check(b._L.ocean_build_mips(b._h, 0), "ocean_build_mips")      # enqueued behind it at once: consumes whatever the frame leaves
try:
    b.synchronize()
    print("5. consumer behind a faulted frame: NO error reported: FAIL")
    ok = False
except W.OceanError as e:
    print(f"5. consumer behind a faulted frame: reported once ({e.code}, hip {W._abi.last_hip_error()}); ", end="")
b.synchronize()                                 # the context is usable, nothing is sticky
m = b.build_mips(0)                             # the caller repeats the consumer call: the recovered frame's mips
r = W.OceanBatch(256, 1, 0); r.set_merged_xpass(False); r.prepare(7); r.compute_waves(0.5); mr = r.build_mips(0); r.close()
right = all(np.array_equal(x, y) for lv, lr in zip(m, mr) for x, y in zip(lv, lr))
print(f"repeated consumer call right: {right}, recoveries {b.fault_recoveries}")
ok &= right and b.fault_recoveries == 1
b.close()
print("FAULT_PATH_OK" if ok else "FAULT_PATH_FAIL")
sys.exit(0 if ok else 1)
