"""bench.py's sequence on the main context with a varying number of pipelined frames before the serial per-kernel pass (developer tool)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import watersurfacerendering_amd as W
for frames in [int(x) for x in sys.argv[1].split(",")]:
    b = W.OceanBatch(2048, 1, 0); b.set_pipeline_depth(3); b.prepare(0x5EED0000)
    for j in range(frames): b.compute_waves_async(0.05 * j)
    b.synchronize()
    _, kp = b.time_frames(0.0, 0.05, 200, 200)
    b.set_pipeline_depth(1)
    ms, k = b.time_frames(0.0, 0.05, 200, 200)
    ms2, k2 = b.time_frames(0.0, 0.05, 200, 200)
    print(f"{frames} pipelined frames first: serial {ms/200*1e3:6.1f} us  " + "  ".join(f"{v*1e3:6.2f}" for v in k) + f"   again {ms2/200*1e3:6.1f}  " + "  ".join(f"{v*1e3:6.2f}" for v in k2), flush=True)
    b.close()
