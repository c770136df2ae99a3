"""Is k_xpass_b's serial duration bimodal from context to context?  (developer tool)  usage: bimodal.py N contexts"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import watersurfacerendering_amd as W
n = int(sys.argv[1]); reps = int(sys.argv[2])
keep = []
for r in range(reps):
    b = W.OceanBatch(n, 1, 0)
    if os.environ.get("OCEAN_DEPTH"): b.set_pipeline_depth(int(os.environ["OCEAN_DEPTH"]))
    b.prepare(0x5EED0000 + r)
    if os.environ.get("OCEAN_DEPTH"):
        b.time_frames(0.0, 0.05, 100, 100, per_kernel=False); b.set_pipeline_depth(1)
    b.time_frames(0.0, 0.05, 200, 50, per_kernel=False)
    ms, k = b.time_frames(0.0, 0.05, 100, 200)
    pd, pq = ctypes.c_void_p(), ctypes.c_void_p()
    b._L.ocean_device_maps(b._h, ctypes.byref(pd), ctypes.byref(pq))
    print(f"ctx {r}: serial {ms/200*1e3:6.1f} us  " + "  ".join(f"{nm} {v*1e3:6.2f}" for nm, v in zip(b.kernel_names(), k)) + f"   disp {pd.value:#x} nrm {pq.value:#x}", flush=True)
    if os.environ.get("OCEAN_KEEP"): keep.append(b)
    else: b.close()
