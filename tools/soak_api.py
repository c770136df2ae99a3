"""Soak of the frame path's host side (round 3): thousands of randomly mixed operations -- synchronous calls (tracked: completion records),
asynchronous frames with and without tracking, ocean_wait_frame, read-outs, mode / depth / lambda changes, resizes -- on one long-lived context,
every returned amplitude and (sampled) map checked against a second context that only ever runs fully synchronised serial frames.
usage: soak_api.py [operations] [seed]"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import watersurfacerendering_amd as W
ops = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
sizes = [64, 256, 512, 1024]
n, tiles = 256, 2
b = W.OceanBatch(n, tiles, 0); r = W.OceanBatch(n, tiles, 0)
state = dict(mode=0, depth=1, lam=-1.0, seed=5)
def reprepare():
    for c in (b, r):
        c.set_mode(state["mode"]); c.set_lambda(state["lam"]); c.prepare(state["seed"])
    b.set_pipeline_depth(state["depth"])
reprepare()
last_t, checked, bad = None, 0, 0
def ref(t):
    r.compute_waves_async(t); r.synchronize()
    return np.array([r.heights(i)[0] for i in range(tiles)], dtype=np.float32)
for k in range(ops):
    op = rng.random()
    t = round(rng.uniform(0.0, 50.0), 3)
    if op < 0.35:
        got = b.compute_waves(t); last_t = t
        bad += not np.array_equal(got, ref(t)); checked += 1
    elif op < 0.60:
        b.compute_waves_async(t); last_t = t
        if rng.random() < 0.5:
            got = b.wait_frame()
            bad += not np.array_equal(got, ref(t)); checked += 1
    elif op < 0.70 and last_t is not None:
        d, q = b.read_maps(tiles - 1, 1)
        ref(last_t); d2, q2 = r.read_maps(tiles - 1, 1)
        bad += not (np.array_equal(d, d2) and np.array_equal(q, q2)); checked += 1
    elif op < 0.75:
        b.set_frame_tracking(rng.random() < 0.5)
    elif op < 0.80:
        state["depth"] = rng.choice([1, 2, 3, 5]); b.set_pipeline_depth(state["depth"])
    elif op < 0.85:
        state["lam"] = rng.choice([-1.0, -0.5, -2.0]); b.set_lambda(state["lam"]); r.set_lambda(state["lam"]); last_t = None   # (takes effect at the next frame)
    elif op < 0.90:
        state["mode"] = rng.choice([0, 0, 3, 1, 2]); b.set_mode(state["mode"]); r.set_mode(state["mode"]); last_t = None
    elif op < 0.92:
        state["seed"] = rng.randrange(1 << 30); reprepare(); last_t = None
    elif op < 0.93:
        n = rng.choice(sizes); b.set_tile_size(n); r.set_tile_size(n); reprepare(); last_t = None
    elif last_t is not None:
        h = [b.heights(i) for i in range(tiles)]
        ref(last_t)
        bad += h != [r.heights(i) for i in range(tiles)]; checked += 1
    if bad:
        print("MISMATCH at operation", k, "state", state, "n", n); break
b.synchronize()
print(f"soak_api: {ops} operations, {checked} checks, {'all identical' if not bad else 'FAILED'}")
b.close(); r.close()
sys.exit(1 if bad else 0)
