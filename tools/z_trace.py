"""The z pass from inside (diagnostic build -DOCEAN_XB_TRACE; profiles/r03_xpass_trace.txt section 5): start / end / CU of every workgroup of a serial
2048^2 k_zpass (1025 columns, the last round's columns split over two workgroups each).  Prints the schedule per CU: how many workgroups a CU
runs, when its last one ends, how long the tail of the launch is.
The committed -DOCEAN_XB_TRACE build traces the two x passes only; for this script add `XbTrace z_trace_(1024);` behind `const int tile = blockIdx.y;`
at the top of k_zpass (and move the XbTrace definition above it) in a scratch copy -- three lines, kept out of the tree so that the hash the committed
profiles are keyed on does not move for a diagnostic.
usage: OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_xbt.so python tools/z_trace.py [frames]"""
import ctypes as C, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import watersurfacerendering_amd as W
from watersurfacerendering_amd import _abi
L = _abi.lib()
L.ocean_debug_xb_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
b = W.OceanBatch(2048, 1, 0); b.prepare(1)
ms, k = b.time_frames(0.0, 0.05, 200, 300)
print(f"z {k[0]*1e3:.2f} xb {k[1]*1e3:.2f} disp {k[2]*1e3:.2f} us", flush=True)
assert L.ocean_debug_xb_trace(b._h, 1, None, 0) == 0
G = b.last_launch()[0]["grid_x"]
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for j in range(3): b.compute_waves_async(0.1 * j)
    b.synchronize()
    out = np.zeros((1024 + G) * 4, dtype=np.uint64)
    assert L.ocean_debug_xb_trace(b._h, 0, out.ctypes.data_as(C.c_void_p), out.size) == 0
    r = out.reshape(-1, 4).astype(np.int64)[1024:1024 + G]
    t0 = r[:, 0].min()
    start, end = (r[:, 0] - t0) * 0.01, (r[:, 1] - t0) * 0.01
    dur = end - start
    hw, xcc = r[:, 2], r[:, 3] & 15
    cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    per = defaultdict(list)
    for i in range(G): per[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]))].append(i)
    zfull = b.last_launch()[0]["grid_x"] - 2 * (G - 1025) if G > 1025 else G     # whole-column workgroups
    nsplit = G - zfull
    cu_end = np.array([max(end[i] for i in v) for v in per.values()])
    cu_n = np.array([len(v) for v in per.values()])
    cu_busy = np.array([sum(dur[i] for i in v) for v in per.values()])
    print(f"frame {rep}: grid {G} ({zfull} whole columns + {nsplit} half jobs), span {end.max():.2f} us; first-wave starts within {np.sort(start)[min(767, G-1)]:.2f} us")
    print(f"   workgroups per CU: min {cu_n.min()} mean {cu_n.mean():.2f} max {cu_n.max()};  CU end times: p0 {cu_end.min():.2f} p25 {np.percentile(cu_end,25):.2f} p50 {np.percentile(cu_end,50):.2f} p75 {np.percentile(cu_end,75):.2f} p100 {cu_end.max():.2f}")
    print(f"   duration of whole-column workgroups: first round (started < 1 us) mean {dur[:zfull][start[:zfull] < 1].mean():.2f}, later mean {dur[:zfull][start[:zfull] >= 1].mean():.2f} max {dur[:zfull].max():.2f};"
          + (f" half jobs mean {dur[zfull:].mean():.2f} max {dur[zfull:].max():.2f}, start p50 {np.percentile(start[zfull:],50):.2f}" if nsplit else ""))
    idle = (end.max() - cu_end)
    print(f"   idle CU time at the tail: mean {idle.mean():.2f} us per CU ({100*idle.mean()/end.max():.1f} % of the span); start times of the last 10 workgroups: {[round(float(x),1) for x in np.sort(start)[-10:]]}")
    order = np.argsort(-end)[:6]
    print("   last to finish:", [(int(i), round(float(start[i]),1), round(float(dur[i]),1), round(float(end[i]),1), len(per[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]))])) for i in order])
b.close()
