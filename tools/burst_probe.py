import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: F401  (OCEAN_HIP_LIB -> variant library, developer A/B only)
import watersurfacerendering_amd as W
DEPTH = int(sys.argv[1]) if len(sys.argv) > 1 else 3
b = W.OceanBatch(2048, 1, 0); b.set_pipeline_depth(DEPTH); b.prepare(1)
print("depth", DEPTH)
DT=0.016
for j in range(500): b.compute_waves_async(DT*j)
b.synchronize(); torch.cuda.synchronize()
def T(f, n=50):
    xs=[]
    for _ in range(n):
        t=time.perf_counter(); f(); xs.append((time.perf_counter()-t)*1e6)
    xs.sort(); return xs[len(xs)//2]
print("idle b.synchronize us", T(b.synchronize))
print("idle torch.cuda.synchronize us", T(torch.cuda.synchronize))
for K in (20, 21, 24, 100, 1000):
    rows=[]
    for rep in range(15):
        b.synchronize(); torch.cuda.synchronize()
        t0=time.perf_counter()
        for j in range(K): b.compute_waves_async(DT*j)
        t1=time.perf_counter()
        b.synchronize()
        t2=time.perf_counter()
        torch.cuda.synchronize()
        t3=time.perf_counter()
        rows.append(((t1-t0)*1e6,(t2-t0)*1e6,(t3-t0)*1e6))
    rows.sort(key=lambda r:r[2]); m=rows[len(rows)//2]
    print(f"K={K}: enqueue done {m[0]:.0f} us, b.sync done {m[1]:.0f}, torch sync done {m[2]:.0f} -> {m[2]/K:.2f} us/frame; best {rows[0][2]/K:.2f}")
