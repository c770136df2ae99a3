"""What ends the slow-k_xpass_b state of a process?  (profiles/r03_bimodal_probe.txt section 5.)  About one process in four runs serial k_xpass_b at
22.5-24 instead of 21 us for its whole life.  This script times a first context; if the process turns out slow (or with 'always') it tries, one
after the other, each followed by a fresh context: new allocations, a 6 GiB blocker allocation ahead of the context, 20 extra HIP streams, two
seconds of idleness, frames on the context's other pipeline chains (other streams = other hardware queues).
usage: xb_interventions.py [always]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
T0 = time.perf_counter()
import watersurfacerendering_amd as W
hip = C.CDLL("libamdhip64.so")

def ctx_times(tag, keep=None):
    b = W.OceanBatch(2048, 1, 0); b.prepare(0x5EED0000)
    b.time_frames(0.0, 0.05, 200, 50, per_kernel=False)
    ms, k = b.time_frames(0.0, 0.05, 100, 300)
    print(f"{time.perf_counter() - T0:6.2f} s {tag:34s} z {k[0]*1e3:6.2f} xb {k[1]*1e3:6.2f} disp {k[2]*1e3:6.2f} frame {ms/300*1e3:6.1f}", flush=True)
    if keep is not None: keep.append(b)
    else: b.close()
    return k[1] * 1e3

first = ctx_times("first context")
if first <= 22.4 and "always" not in sys.argv:
    print("normal process"); sys.exit(0)
print("SLOW process" if first > 22.4 else "normal process, interventions anyway")
for i in range(2): ctx_times(f"fresh context {i}")
blk = C.c_void_p(); rc = hip.hipMalloc(C.byref(blk), C.c_size_t(6 << 30))
ctx_times(f"behind a 6 GiB blocker (rc {rc})")
blk2 = C.c_void_p(); rc = hip.hipMalloc(C.byref(blk2), C.c_size_t(40 << 30))
ctx_times(f"behind 46 GiB of blockers (rc {rc})")
hip.hipFree(blk); hip.hipFree(blk2)
ctx_times("blockers freed")
streams = []
for i in range(20):
    s = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(s), 1); streams.append(s)
ctx_times("after 20 extra streams")
time.sleep(2.0)
ctx_times("after 2 s idle")
# the other chains of ONE context: depth 4 puts consecutive frames on streams 0..3; serial timing per chain is not exposed, so time whole frames
b = W.OceanBatch(2048, 1, 0); b.prepare(0x5EED0000)
for depth in (1, 2, 3, 4):
    b.set_pipeline_depth(depth)
    ms, _ = b.time_frames(0.0, 0.05, 300, 600, per_kernel=False)
    print(f"{time.perf_counter() - T0:6.2f} s pipelined depth {depth}: {ms/600*1e3:6.1f} us/frame", flush=True)
b.close()
ctx_times("last context")
