"""Where and when the 387 workgroups of a serial 2048^2 k_xpass_b run (diagnostic build: make -C watersurfacerendering_amd/csrc variant NAME=xbtrace
DEFS=-DOCEAN_XB_TRACE; OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_xbtrace.so).  Per workgroup: start / end on the 100 MHz wall clock,
HW_REG_HW_ID (CU, shader array, shader engine, pipe, queue, micro-engine, VMID) and HW_REG_XCC_ID.  Prints the queue the kernel came through, how
the HEIGHT (blocks 0..129) and NORMAL (130..386) workgroups were paired on the CUs, and their durations.
usage: xb_trace.py [none | stream_before | two_streams_before] [frames]"""
import ctypes as C, os, sys
from collections import Counter, defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
if mode != "none":
    import torch
    keep = [torch.cuda.Stream() for _ in range(2 if mode == "two_streams_before" else 1)]
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: F401  (OCEAN_HIP_LIB -> variant library, developer A/B only)
import watersurfacerendering_amd as W
from watersurfacerendering_amd import _abi
L = _abi.lib()
L.ocean_debug_xb_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
b = W.OceanBatch(2048, 1, 0); b.prepare(1)
ms, k = b.time_frames(0.0, 0.05, 200, 300)
print(f"[{mode}] z {k[0]*1e3:.2f} xb {k[1]*1e3:.2f} disp {k[2]*1e3:.2f} us  -> {'SLOW' if k[1]*1e3 > 22.4 else 'normal'}", flush=True)
assert L.ocean_debug_xb_trace(b._h, 1, None, 0) == 0
G, HB, NBD = 387, 130, 257
def analyse(name, r, nh, t0=None):
    """r: [blocks][4] records; the first nh blocks are HEIGHT workgroups (0 for the displacement pass)"""
    g = len(r)
    t0 = r[:, 0].min()
    start, end = (r[:, 0] - t0) * 0.01, (r[:, 1] - t0) * 0.01          # us
    hw, xcc = r[:, 2], r[:, 3] & 15
    cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    where = defaultdict(list)
    for i in range(g): where[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]))].append(i)
    role = lambda i: "H" if i < nh else "N"
    pairs = Counter("+".join(sorted(role(i) for i in v)) for v in where.values())
    dur = end - start
    order = np.argsort(-end)[:4]
    last = [(int(i), role(i), round(float(dur[i]), 2), round(float(end[i]), 2), [(j, round(float(dur[j]), 2)) for j in where[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]))] if j != i]) for i in order]
    print(f"  {name}: span {end.max():6.2f} us, starts within {start.max():4.2f}; N duration mean {dur[nh:].mean():5.2f} p90 {np.percentile(dur[nh:], 90):5.2f} max {dur[nh:].max():5.2f}"
          + (f"; H mean {dur[:nh].mean():5.2f} max {dur[:nh].max():5.2f}" if nh else "") + f"; CUs {len(where)} pairing {dict(pairs)}")
    print(f"     last to finish (block, role, duration, end, CU mates): {last}")
    return end.max()
spans = defaultdict(list)
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    for j in range(3): b.compute_waves_async(0.1 * j)
    b.synchronize()
    out = np.zeros((512 + NBD) * 4, dtype=np.uint64)
    assert L.ocean_debug_xb_trace(b._h, 0, out.ctypes.data_as(C.c_void_p), out.size) == 0
    r = out.reshape(-1, 4).astype(np.int64)
    if os.environ.get("XB_TRACE_SAVE"):
        np.save(f"{os.environ['XB_TRACE_SAVE']}_{'slow' if k[1]*1e3 > 22.4 else 'normal'}_{os.getpid()}_{rep}.npy", r)
    spans["xb"].append(analyse("k_xpass_b   ", r[:G], HB))
    spans["xd"].append(analyse("k_xpass_disp", r[512:512 + NBD], 0))
print("  mean spans:", {kk: round(float(np.mean(v)), 2) for kk, v in spans.items()})
b.close()
