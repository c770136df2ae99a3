"""Per-kernel execution times (dispatch-attached events) of serial frames, for A/B runs over variant libraries.
usage: OCEAN_HIP_LIB=... python tools/kernel_times.py N [tiles] [frames]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: F401  (OCEAN_HIP_LIB -> variant library, developer A/B only)
import watersurfacerendering_amd as W
n = int(sys.argv[1]); tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 1; frames = int(sys.argv[3]) if len(sys.argv) > 3 else 300
b = W.OceanBatch(n, tiles, 0)
if os.environ.get('OCEAN_Z16'): b.set_intermediate_precision(16)
b.prepare(0x5EED0000)
b.time_frames(0.0, 0.05, 200, 50, per_kernel=False)
ms, k = b.time_frames(0.0, 0.05, 100, frames)
print(f"N={n}x{tiles} serial {ms/frames*1e3:7.1f} us/frame   " + "  ".join(f"{nm} {v*1e3:6.2f}" for nm, v in zip(b.kernel_names(), k)))
b.close()
