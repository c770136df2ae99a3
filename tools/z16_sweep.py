"""Randomised sweep of the half2-intermediate mode against the fp32 path of the same library (developer tool):
many seeds, parameter sets, sizes and times; prints the worst error per channel relative to the channel's maximum."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import watersurfacerendering_amd as W
rng = np.random.default_rng(2)
worst = 0.0; bad = 0; cases = 0
for n in (32, 64, 128, 512):
    for _ in range(12 if n <= 128 else 4):
        seed = int(rng.integers(1, 1 << 30))
        p = dict(tile_length=float(rng.choice([100.0, 250.0, 1000.0, 4000.0])), wind_dir_x=float(rng.uniform(-1, 1)), wind_dir_y=float(rng.uniform(0.05, 1)),
                 wind_speed=float(rng.uniform(2, 60)), phillips_const=float(10 ** rng.uniform(-8, -5)), damping=float(rng.uniform(0.0, 1.0)), lambda_=float(rng.uniform(-3, -0.2)))
        mode = int(rng.choice([0, 3]))
        a = W.OceanBatch(n, 1, 0); a.set_params(**p); a.set_mode(mode); a.prepare(seed)
        b = W.OceanBatch(n, 1, 0); b.set_params(**p); b.set_mode(mode); b.set_intermediate_precision(16); b.prepare(seed)
        for t in (0.0, float(rng.uniform(0, 500))):
            a.compute_waves(t); b.compute_waves(t)
            d1, q1 = a.read_maps(); d2, q2 = b.read_maps()
            cases += 1
            if not (np.all(np.isfinite(d2)) and np.all(np.isfinite(q2))):
                bad += 1; print("NON-FINITE", n, seed, p, mode, t); continue
            for x, y in ((d1, d2), (q1, q2)):
                for c in range(4):
                    m = float(np.abs(x[..., c]).max())
                    if m > 0: worst = max(worst, float(np.abs(x[..., c] - y[..., c]).max()) / m)
        a.close(); b.close()
print(f"z16 sweep: {cases} frames, {bad} non-finite, worst error {worst:.2e} of a channel's maximum")
