import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: F401  (OCEAN_HIP_LIB -> variant library, developer A/B only)
import watersurfacerendering_amd as W
n = int(sys.argv[1]); tiles = int(sys.argv[2]); depths = [int(x) for x in sys.argv[3].split(",")]
out = []
for depth in depths:
    b = W.OceanBatch(n, tiles, 0)
    if os.environ.get('OCEAN_FP16'): b.set_spectrum_precision(16)
    if os.environ.get('OCEAN_Z16'): b.set_intermediate_precision(16)
    if os.environ.get('OCEAN_MODE'): b.set_mode(int(os.environ['OCEAN_MODE']))
    b.prepare(1); b.set_pipeline_depth(depth)
    frames = int(os.environ.get("OCEAN_FRAMES", "1000"))
    ms, _ = b.time_frames(0.0, 0.05, int(os.environ.get("OCEAN_WARMUP", "500")), frames, per_kernel=False)
    per = ms / frames * 1e3
    out.append(f"d{depth}={per:.1f}us ({n*n*tiles/per/1e3:.1f} Gtexel/s)")
    b.close()
print(f"N={n} tiles={tiles} " + " ".join(out))
