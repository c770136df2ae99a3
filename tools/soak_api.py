"""Soak of the frame path's host side (round 3): thousands of randomly mixed operations -- synchronous calls (tracked: completion records),
asynchronous frames with and without tracking, ocean_wait_frame, ocean_compute_waves_read, read-outs (plain, asynchronous into registered memory, staging layout), mode /
depth / lambda / parameter / dispersion / precision / time-offset changes, resizes, mip chains, the vertex-stage consumer, dma-buf exports -- on
one long-lived context, every returned amplitude and (sampled) map checked against a second context that only ever runs fully synchronised
serial frames and is re-created from scratch now and then (so that stale state in the long-lived one cannot hide in both).
usage: soak_api.py [operations] [seed]"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: F401  (OCEAN_HIP_LIB -> variant library, developer A/B only)
import watersurfacerendering_amd as W
ops = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
sizes = [int(x) for x in os.environ.get("SOAK_SIZES", "64,256,512,1024").split(",")]
n, tiles = 256, 2
b = W.OceanBatch(n, tiles, 0); r = W.OceanBatch(n, tiles, 0)
state = dict(mode=0, depth=1, lam=-1.0, seed=5, inter=32, spec=32, disp=(0, 0.0), wind=(20.0, 1.0, 0.5), length=1000.0, offs=None)
def reprepare():
    for c in (b, r):
        c.set_mode(state["mode"]); c.set_intermediate_precision(state["inter"]); c.set_spectrum_precision(state["spec"])
        c.set_dispersion(*state["disp"])
        c.set_params(wind_speed=state["wind"][0], wind_dir_x=state["wind"][1], wind_dir_y=state["wind"][2], tile_length=state["length"])
        c.set_lambda(state["lam"]); c.set_time_offsets(state["offs"]); c.prepare(state["seed"])
    b.set_pipeline_depth(state["depth"])
def fresh_reference():
    """the reference context from scratch: whatever the long-lived one has been through, a new one must agree with it"""
    global r
    r.close(); r = W.OceanBatch(n, tiles, 0)
    r.set_mode(state["mode"]); r.set_intermediate_precision(state["inter"]); r.set_spectrum_precision(state["spec"]); r.set_dispersion(*state["disp"])
    r.set_params(wind_speed=state["wind"][0], wind_dir_x=state["wind"][1], wind_dir_y=state["wind"][2], tile_length=state["length"])
    r.set_lambda(state["lam"]); r.set_time_offsets(state["offs"]); r.prepare(state["seed"])
reprepare()
last_t, checked, bad = None, 0, 0
trace = bool(os.environ.get("SOAK_TRACE"))
import ctypes as C
_hip = C.CDLL("libamdhip64.so"); kept = []; pool = {}
skip = set(os.environ.get("SOAK_SKIP", "").split(","))      # debugging: leave out "read", "merge", "search", "pin"
def ref(t):
    r.compute_waves_async(t); r.synchronize()
    return np.array([r.heights(i)[0] for i in range(tiles)], dtype=np.float32)
for k in range(ops):
    op = rng.random()
    t = round(rng.uniform(0.0, 50.0), 3)
    if trace: print(f"op {k} {op:.4f} n={n} depth={state['depth']} mode={state['mode']}", file=sys.stderr, flush=True)
    if op < 0.29:
        got = b.compute_waves(t); last_t = t
        bad += not np.array_equal(got, ref(t)); checked += 1
    elif op < 0.35 and "read" not in skip:
        # round 6: the one-call frame + read-out (direct host stores for small page-locked destinations, copies otherwise)
        pinned = rng.random() < 0.5 and "pin" not in skip
        persistent = pinned and "churn" not in skip
        if persistent:              # the adaptor's pattern (include/WSTessendorf.hpp): destinations registered once per size and kept
            if n not in pool:
                pool[n] = (np.empty((tiles, n, n, 4), np.float32), np.empty((tiles, n, n, 4), np.float32))
                for a in pool[n]: W.host_register(a)
            d, q = pool[n]
        else:
            d = np.empty((tiles, n, n, 4), np.float32); q = np.empty_like(d)
        if trace: print(f"   read pinned={pinned} d={d.ctypes.data:#x} q={q.ctypes.data:#x} bytes={d.nbytes}", file=sys.stderr, flush=True)
        if pinned and "rawreg" in skip:      # debugging: the runtime's calls directly, not ocean_host_register
            for a in (d, q): assert _hip.hipHostRegister(C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 0) == 0
        elif pinned and not persistent: W.host_register(d); W.host_register(q)
        if trace: print("   registered", file=sys.stderr, flush=True)
        if "oldapi" in skip:      # debugging: the same registered destinations through round 5's calls
            got = b.compute_waves(t); b.read_maps_async(d, q); b.synchronize(); last_t = t
        else:
            got, _, _ = b.compute_waves_read(t, d, q); last_t = t
        if trace: print("   call returned", file=sys.stderr, flush=True)
        if pinned and "rawreg" in skip:
            for a in (d, q): assert _hip.hipHostUnregister(C.c_void_p(a.ctypes.data)) == 0
        elif pinned and not persistent: W.host_unregister(d); W.host_unregister(q)
        if pinned and "keep" in skip: kept.append((d, q))      # debugging: formerly registered arrays are never freed
        if trace: print("   unregistered", file=sys.stderr, flush=True)
        want = ref(t)
        if trace: print("   reference frame done", file=sys.stderr, flush=True)
        d2, q2 = r.read_maps()
        if trace: print(f"   reference read-out done d2={d2.ctypes.data:#x} q2={q2.ctypes.data:#x}", file=sys.stderr, flush=True)
        bad += not (np.array_equal(got, want) and np.array_equal(d, d2) and np.array_equal(q, q2)); checked += 1
    elif op < 0.60:
        b.compute_waves_async(t); last_t = t
        if rng.random() < 0.5:
            got = b.wait_frame()
            bad += not np.array_equal(got, ref(t)); checked += 1
    elif op < 0.70 and last_t is not None:
        d, q = b.read_maps(tiles - 1, 1)
        ref(last_t); d2, q2 = r.read_maps(tiles - 1, 1)
        bad += not (np.array_equal(d, d2) and np.array_equal(q, q2)); checked += 1
    elif op < 0.74:
        b.set_frame_tracking(rng.random() < 0.5)
    elif op < 0.75 and "merge" not in skip:
        b.set_merged_xpass(rng.random() < 0.7)       # (round 5/6: in-launch hand-offs on / off; same bits)
    elif op < 0.80:
        state["depth"] = rng.choice([1, 2, 3, 5]); b.set_pipeline_depth(state["depth"])
    elif op < 0.85:
        state["lam"] = rng.choice([-1.0, -0.5, -2.0]); b.set_lambda(state["lam"]); r.set_lambda(state["lam"]); last_t = None   # (takes effect at the next frame)
    elif op < 0.90:
        state["mode"] = rng.choice([0, 0, 3, 1, 2]); b.set_mode(state["mode"]); r.set_mode(state["mode"]); last_t = None
    elif op < 0.92 and "reprepare" not in skip:
        state["seed"] = rng.randrange(1 << 30); reprepare(); last_t = None
    elif op < 0.93 and "resize" not in skip:
        n = rng.choice(sizes); b.set_tile_size(n); r.set_tile_size(n)
        b.set_placement_search(1 if "search" in skip else rng.choice([0, 1, 3])); reprepare(); last_t = None      # (round 6: Prepare's placement search on / off / forced at any size)
    elif op < 0.94 and "reprepare" not in skip:
        what = rng.randrange(6)
        if what == 0: state["inter"] = rng.choice([16, 32])
        elif what == 1: state["spec"] = rng.choice([16, 32])
        elif what == 2: state["disp"] = rng.choice([(0, 0.0), (1, 25.0), (2, 0.05)])
        elif what == 3: state["wind"] = (rng.uniform(2.0, 40.0), rng.uniform(-1, 1), rng.uniform(0.1, 1))
        elif what == 4: state["length"] = rng.choice([250.0, 1000.0, 3000.0])
        else: state["offs"] = rng.choice([None, [0.25 * i for i in range(tiles)]])
        reprepare(); last_t = None
    elif op < 0.945 and "fresh" not in skip:
        fresh_reference()
    elif op < 0.955 and last_t is not None:
        tile = rng.randrange(tiles)
        md, mq = b.build_mips(tile); ref(last_t); rd, rq = r.build_mips(tile)
        bad += not all(np.array_equal(x, y) for x, y in zip(md + mq, rd + rq)); checked += 1
    elif op < 0.965 and last_t is not None:
        tile, g = rng.randrange(tiles), rng.choice([16, 64])
        p1, q1 = b.displace_grid(tile, g, uv_scale=rng.choice([1.0, 0.37])); 
        ref(last_t); p2, q2 = r.displace_grid(tile, g, uv_scale=1.0)
        p3, q3 = b.displace_grid(tile, g, uv_scale=1.0)
        bad += not (np.array_equal(p3, p2) and np.array_equal(q3, q2)); checked += 1
    elif op < 0.975 and last_t is not None:
        tile = rng.randrange(tiles)
        vb, ib = rng.choice([(0, 0), (40, 12), (4096, 36)])
        off = (vb + ib + 15) // 16 * 16
        stag = np.zeros(off + 2 * n * n * 16, dtype=np.uint8)
        used = b.read_maps_staging(stag, vb, ib, tile); b.synchronize()
        ref(last_t); d2, q2 = r.read_maps(tile, 1)
        bad += not (used == stag.size and stag[off:].tobytes() == d2.tobytes() + q2.tobytes()); checked += 1
    elif op < 0.98 and last_t is not None:
        d = np.empty((tiles, n, n, 4), np.float32); q = np.empty_like(d)
        b.read_maps_async(d, q); b.synchronize()
        ref(last_t); d2, q2 = r.read_maps()
        bad += not (np.array_equal(d, d2) and np.array_equal(q, q2)); checked += 1
    elif op < 0.985 and last_t is not None and state["depth"] == 1 and "export" not in skip:
        fd = b.export_maps()[0]; os.close(fd)
    elif op < 0.99 and "select" not in skip:
        b.select_streams(3); last_t = None          # streams re-ordered; the maps hold a calibration frame
    elif last_t is not None:
        h = [b.heights(i) for i in range(tiles)]
        ref(last_t)
        bad += h != [r.heights(i) for i in range(tiles)]; checked += 1
    if bad:
        print("MISMATCH at operation", k, "state", state, "n", n); break
b.synchronize()
for pair in pool.values():
    for a in pair: W.host_unregister(a)
print(f"soak_api: {ops} operations, {checked} checks, {'all identical' if not bad else 'FAILED'}")
b.close(); r.close()
sys.exit(1 if bad else 0)
