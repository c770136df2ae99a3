"""GPU parity in the regimes bench.py actually times (all through the C ABI):

  * 2048^2 at pipeline depth 3 and 4096^2 at depth 2 -- the non-temporal map-store instantiations
    (`k_xpass_b/k_xpass_disp<..., NTS = true>`) and, at 4096^2, the streamed intermediates (`ZNT`);
  * BASELINE config 5's per-GPU share: 8 x 1024^2 tiles per launch, seeds 0x5EED0000 + i, at depth 1 and at
    depth 2 (which selects `store_z<ZNT = true>`), every tile against the oracle, plus the committed
    fixture tests/golden/sampled_n1024_default.npz through the batched path;
  * 2048^2 / 4096^2 with the alternative parameter set at t = 0, 1000 and 3e5 s (the last one takes the
    library sincos path: omega*t > 1e5 rad);
  * a caller-owned stream (ocean_set_stream).

Tolerances as in tests/test_parity_gpu.py: per channel max|err| <= 1e-5 * max|channel| against the oracle with
float64 FFTs, amplitude 1e-6 relative; pipelined frames must equal serial frames bit for bit.
Reference lines: WSTessendorf.cpp:284-455.
"""
import os

import numpy as np
from conftest import pinned_array
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-5
TOL_AMP = 1e-6
SEED = 0x5EED0000
ALT = dict(length=250.0, wind=(1.0, 0.0), wind_speed=10.0, lam=-2.0)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def chan_err(a, b):
    out = []
    for c in range(4):
        den = max(float(np.abs(b[..., c]).max()), 1e-30)
        out.append(float(np.abs(a[..., c].astype(np.float64) - b[..., c]).max()) / den)
    return out


def make_oracle(n, xi, length=1000.0, **kw):
    from oracle import oracle as O
    o = O.Oracle(n, length, **kw)
    o.prepare(xi=xi)
    return o


def make_batch(n, tiles=1, seed=SEED, depth=1, length=1000.0, wind=(1.0, 1.0), wind_speed=30.0, lam=-1.0):
    import watersurfacerendering_amd as W
    b = W.OceanBatch(n, tiles, 0)
    b.set_params(tile_length=length, wind_dir_x=wind[0], wind_dir_y=wind[1], wind_speed=wind_speed, lambda_=lam)
    b.set_pipeline_depth(depth)
    b.prepare(seed)
    return b


def assert_matches_oracle(d, q, heights, o, t, what):
    from oracle import oracle as O
    ao, do, no = o.compute_waves(t, fft=O.FFT_F64)
    a, mn, mx = heights
    assert abs(a - ao) <= TOL_AMP * abs(ao), (what, a, ao)
    assert abs(mn - o.min_height) <= TOL_AMP * abs(ao) and abs(mx - o.max_height) <= TOL_AMP * abs(ao), what
    ed, en = chan_err(d, do), chan_err(q, no)
    assert max(ed) <= TOL, (what, "displacement", ed)
    assert max(en) <= TOL, (what, "normal", en)
    assert np.all(d[..., 3] == 1.0)


@pytest.mark.parametrize("n,depth", [(2048, 3), (4096, 2)])
def test_pipelined_headline_regime_matches_serial_and_oracle(n, depth):
    """The regime `bench.py` times: asynchronous frames rotating over `depth` chains (maps stored non-temporally).
    Sequences of depth, depth+1, ... frames end on every chain in turn; the last frame of each must equal the
    serial context's frame bit for bit and the float64 oracle within 1e-5."""
    ser = make_batch(n, depth=1)
    pip = make_batch(n, depth=depth)
    o = make_oracle(n, ser.read_xi(0))
    j = 0
    for extra in range(depth):
        times = [0.05 * (j + i) for i in range(depth + extra)]
        j += len(times)
        for t in times:
            pip.compute_waves_async(t)
        pip.synchronize()
        ser.compute_waves(times[-1])
        d1, q1 = ser.read_maps(); d2, q2 = pip.read_maps()
        assert np.array_equal(d1, d2) and np.array_equal(q1, q2), (n, depth, extra)
        assert ser.heights(0) == pip.heights(0)
        if extra == 0 or n <= 2048:      # one oracle frame per chain at 2048^2, one in all at 4096^2 (oracle cost)
            assert_matches_oracle(d2[0], q2[0], pip.heights(0), o, times[-1], (n, depth, extra))
    ser.close(); pip.close()


@pytest.mark.parametrize("n", [2048, 4096])
def test_full_size_alt_params_and_times(n):
    """2048^2 / 4096^2 with non-default spectrum parameters at t = 0, 1000 and 3e5 (library sincos path)."""
    b = make_batch(n, seed=SEED + 17, **ALT)
    o = make_oracle(n, b.read_xi(0), **ALT)
    _, om = b.read_spectrum(0)
    assert np.array_equal(om, o.omega)
    assert float(om.max()) * 3.0e5 > 1.0e5                      # the large-argument branch really is taken
    for t in (0.0, 1000.0, 3.0e5):
        b.compute_waves(t)
        d, q = b.read_maps()
        assert_matches_oracle(d[0], q[0], b.heights(0), o, t, (n, t))
    b.close()


def test_config5_share_batch_of_8_tiles_1024():
    """BASELINE config 5, one rank's share: 8 tiles of 1024^2 per launch, seeds 0x5EED0000 + i.  Depth 1 runs the
    plain z-pass stores, depth 2 the streamed ones (`ZNT`) and non-temporal map stores: every tile against the
    oracle at depth 1, depth 2 bit-identical to it on every chain."""
    from oracle import oracle as O
    n, tiles = 1024, 8
    b1 = make_batch(n, tiles=tiles, depth=1)
    b2 = make_batch(n, tiles=tiles, depth=2)
    offs = np.linspace(0.0, 3.5, tiles).astype(np.float32)      # every tile at its own time as well
    b1.set_time_offsets(offs); b2.set_time_offsets(offs)
    t = 1.25
    amps = b1.compute_waves(t)
    d1, q1 = b1.read_maps()
    for i in range(tiles):
        xi = b1.read_xi(i)
        assert np.abs(xi - O.gauss_xi_numpy(SEED + i, n)).max() <= 1e-6 * np.abs(xi).max()    # tile i = seed + i
        o = make_oracle(n, xi)
        assert_matches_oracle(d1[i], q1[i], b1.heights(i), o, float(np.float32(t) + offs[i]), ("tile", i))
        assert float(amps[i]) == b1.heights(i)[0]
    for frames in (2, 3):                                       # last frame on chain 1, then on chain 0
        for k in range(frames - 1):
            b2.compute_waves_async(0.3 * k)
        b2.compute_waves_async(t)
        b2.synchronize()
        d2, q2 = b2.read_maps()
        assert np.array_equal(d1, d2) and np.array_equal(q1, q2), frames
        for i in range(tiles):
            assert b1.heights(i) == b2.heights(i)
    b1.close(); b2.close()


@pytest.mark.parametrize("depth", [1, 2])
def test_committed_1024_fixture_through_the_batched_path(depth):
    """tests/golden/sampled_n1024_default.npz (seed 0x5EED0000 + 1024) as tile 3 of an 8-tile launch."""
    g = np.load(os.path.join(GOLDEN, "sampled_n1024_default.npz"))
    n, tile = int(g["n"]), 3
    assert n == 1024 and int(g["seed"]) == SEED + 1024
    b = make_batch(n, tiles=8, seed=int(g["seed"]) - tile, depth=depth)
    idx = g["index"]
    for i, t in enumerate(g["times"]):
        if depth > 1:
            b.compute_waves_async(float(t) + 0.5)               # keeps the other chain busy with a different frame
        b.compute_waves_async(float(t))
        b.synchronize()
        amp = b.heights(tile)[0]
        d, q = b.read_maps(tile, 1)
        d, q = d[0].reshape(-1, 4), q[0].reshape(-1, 4)
        scale = np.maximum(g[f"maxabs{i}"], 1e-30)
        assert abs(amp - float(g[f"amp{i}"])) <= 2 * TOL_AMP * amp
        mean = np.concatenate([np.abs(d).mean(0, dtype=np.float64), np.abs(q).mean(0, dtype=np.float64)])
        assert np.all(np.abs(mean - g[f"meanabs{i}"]) <= TOL * scale)
        assert np.all(np.abs(d[idx] - g[f"disp{i}"]) <= TOL * scale[:4])
        assert np.all(np.abs(q[idx] - g[f"nrm{i}"]) <= TOL * scale[4:])
    b.close()


def test_caller_stream_matches_internal_stream_and_orders_with_it():
    """ocean_set_stream: frames enqueued on a caller-owned HIP stream (a torch stream here) give the same maps,
    are ordered with the caller's own work on that stream, and NULL restores the internal streams."""
    import torch
    import watersurfacerendering_amd as W
    n = 512
    ref = make_batch(n, seed=77)
    ref.compute_waves(0.75)
    d_ref, q_ref = ref.read_maps()

    b = make_batch(n, seed=77, depth=3)                         # a caller stream also forces serial frames
    s = torch.cuda.Stream(device="cuda:0")
    b.set_stream(s.cuda_stream)
    assert b.stream == s.cuda_stream
    maps = torch.zeros((2, n, n, 4), dtype=torch.float32, device="cuda:0")
    b.bind_output(maps[0].data_ptr(), maps[1].data_ptr())
    with torch.cuda.stream(s):
        maps.fill_(123.0)                                       # caller work before the frame, same stream
        for t in (0.1, 0.4, 0.75):
            b.compute_waves_async(t)
        total = maps.sum(dtype=torch.float64)                   # caller work after the frame: no host sync between
        snap = maps.clone()
    s.synchronize()
    got = snap.cpu().numpy()
    assert np.array_equal(got[0], d_ref[0]) and np.array_equal(got[1], q_ref[0])
    assert float(total) == pytest.approx(float(d_ref.astype(np.float64).sum() + q_ref.astype(np.float64).sum()), rel=1e-9)
    assert b.heights(0) == ref.heights(0)
    # synchronous call on the caller stream, then back to the internal streams
    a = b.compute_waves(0.75)
    assert float(a[0]) == ref.heights(0)[0]
    b.bind_output(None, None)
    b.set_stream(None)
    assert b.stream != s.cuda_stream
    for t in (0.2, 0.75):
        b.compute_waves_async(t)
    b.synchronize()
    d, q = b.read_maps()
    assert np.array_equal(d, d_ref) and np.array_equal(q, q_ref)
    ref.close(); b.close()


def test_setters_do_not_drain_the_device_and_lambda_reaches_the_next_frame():
    """SetLambda / set_params are host-side until the next frame (WSTessendorf.cpp:497-500): per-tile lambdas of a
    batch arrive with one upload at that frame, a uniform lambda by value."""
    from oracle import oracle as O
    n, tiles = 128, 3
    b = make_batch(n, tiles=tiles, seed=5)
    b.compute_waves(1.0)
    d0, q0 = b.read_maps()
    b.set_lambda(-2.0)                                           # uniform
    b.compute_waves(1.0)
    d1, q1 = b.read_maps()
    assert np.array_equal(q0, q1) and np.allclose(d1[..., 0], 2.0 * d0[..., 0], rtol=1e-6, atol=0)
    lams = [-0.5, -1.5, -3.0]
    for i, lam in enumerate(lams):                               # per tile
        b.set_lambda(lam, tile=i)
    for _ in range(2):                                           # second frame: no re-upload, same result
        b.compute_waves(1.0)
        d2, _ = b.read_maps()
        for i, lam in enumerate(lams):
            assert np.allclose(d2[i][..., 0], -lam * d0[i][..., 0], rtol=1e-6, atol=0)
            assert np.allclose(d2[i][..., 2], -lam * d0[i][..., 2], rtol=1e-6, atol=0)
            assert np.array_equal(d2[i][..., 1], d0[i][..., 1])
    # patching one field of every tile keeps each tile's other parameters (lambda here)
    b.set_params(wind_speed=12.0)
    assert [b.get_params(i).lambda_ for i in range(tiles)] == lams
    assert all(b.get_params(i).wind_speed == 12.0 for i in range(tiles))
    b.close()


# ---------------------------------------------------------------------------------------------
# exchange step and upload path
def test_native_rccl_gather_single_rank_and_staging_layout():
    """ocean_gather_maps through RCCL with a one-rank communicator (all a 1-GPU box can form: the gather is then a
    device-local ncclGather), at depth 1 and overlapped at depth 2; and the reference's staging layout
    [vertices | indices | pad16 | displacements | normals] (WaterSurfaceMesh.cpp:701-755) filled by
    ocean_read_maps_staging."""
    import torch
    import watersurfacerendering_amd as W
    n, tiles = 256, 3
    b = make_batch(n, tiles=tiles, seed=91)
    with pytest.raises(W.OceanError):
        b.gather_maps(0, 1, 1)                                  # no communicator yet
    b.comm_init(1, 0, W.comm_unique_id())
    recv = torch.full((2, 1, tiles, n, n, 4), -7.0, dtype=torch.float32, device="cuda:0")
    for depth in (1, 2):
        b.set_pipeline_depth(depth)
        last = None
        for j in range(4):                                      # every gather ordered behind its own frame
            b.compute_waves_async(0.2 * j)
            b.gather_maps(0, recv[0].data_ptr(), recv[1].data_ptr())
            last = 0.2 * j
        b.synchronize()
        d, q = b.read_maps()
        got = recv.cpu().numpy()
        assert np.array_equal(got[0, 0], d) and np.array_equal(got[1, 0], q), depth
        ref = make_batch(n, tiles=tiles, seed=91)
        ref.compute_waves(last)
        d0, q0 = ref.read_maps()
        assert np.array_equal(d, d0) and np.array_equal(q, q0)
        ref.close()
    with pytest.raises(W.OceanError):
        b.gather_maps(1, None, None)                            # root out of range
    # the same gather at half the bytes: the root receives exactly the maps rounded to IEEE half
    recv16 = torch.zeros((2, 1, tiles, n, n, 4), dtype=torch.float16, device="cuda:0")
    b.set_pipeline_depth(2)
    for j in range(3):
        b.compute_waves_async(0.3 * j)
        b.gather_maps(0, recv16[0].data_ptr(), recv16[1].data_ptr(), half=True)
    b.synchronize()
    d, q = b.read_maps()
    got16 = recv16.cpu().numpy()
    assert np.array_equal(got16[0, 0], d.astype(np.float16)) and np.array_equal(got16[1, 0], q.astype(np.float16))
    b.comm_destroy()
    # staging layout, odd mesh sizes so that the pad matters
    vb, ib = 1000 * 32 + 4, 2998 * 4
    off = (vb + ib + 15) // 16 * 16
    staging = pinned_array(off + 2 * n * n * 16 + 64, np.uint8, fill=0xAB)
    b.set_pipeline_depth(1)
    b.compute_waves_async(0.6)
    flush = b.read_maps_staging(staging, vb, ib, tile=1)
    b.synchronize()
    d, q = b.read_maps(1, 1)
    assert flush == off + 2 * n * n * 16
    assert np.all(staging[:off] == 0xAB) and np.all(staging[flush:] == 0xAB)          # mesh data and the tail untouched
    assert np.array_equal(staging[off:off + n * n * 16].view(np.float32).reshape(n, n, 4), d[0])
    assert np.array_equal(staging[off + n * n * 16:flush].view(np.float32).reshape(n, n, 4), q[0])
    b.close()


def test_bench_self_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` (no torchrun, WORLD_SIZE unset) must itself start two rank processes and report
    n_gpus = 2.  On a 1-GPU box the ranks share the device through the developer switch OCEAN_BENCH_BACKEND=gloo
    (RCCL refuses two ranks per device); the driver's multi-GPU runs use the default RCCL backend."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OCEAN_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--prewarm", "20",
                        "--size", "512", "--tiles", "2", "--depth", "2", "--no-extra", "--cpu-seconds", "0.4", "--no-gather"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                       # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["roofline"]["frac"] < 1.0 and out["roofline"]["frame_frac"] < 1.0
    # the N > 1 line keeps the CPU baseline (rank 0's host) and names the sharded workload (config 5's shape: T tiles per GPU)
    assert out["config"]["workload"].startswith("4 x 512x512 tiles, 2 per GPU") and out["config"]["tiles_per_rank"] == 2
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["cores"] >= 1
    assert out["warmup"] == 5 and out["prewarm_frames"] == 20
    # the line is the compact one: a few KB, the detail in the sidecar beside the script
    assert len(lines[0]) < 8192 and out["sidecar"] == "bench_extra.json"
    side = json.load(open(os.path.join(root, out["sidecar"])))
    assert side["line"] == out and side["warmup_frames_effective"] == 25 and side["roofline"]["kernels"]


def test_readout_paths_under_pipelining_refer_to_the_last_frame():
    """ocean_read_maps, ocean_read_maps_async, ocean_read_maps_staging, ocean_device_maps and the vertex-stage consumer
    all refer to the frame enqueued LAST, whatever chain it ran on (depth 3, 1..5 frames in flight)."""
    import watersurfacerendering_amd as W
    n = 256
    ref = make_batch(n, seed=321)
    b = make_batch(n, seed=321, depth=3)
    pinned = pinned_array((2, n, n, 4), np.float32)
    staging = pinned_array(48 + 2 * n * n * 16, np.uint8)
    if True:
        for frames in (1, 2, 3, 4, 5):
            times = [0.11 * (frames * 10 + j) for j in range(frames)]
            for t in times:
                b.compute_waves_async(t)
            b.read_maps_async(pinned[0:1], pinned[1:2])             # enqueued behind the last frame, no host wait
            b.read_maps_staging(staging, 30, 10)                     # maps at align16(40) = 48
            pos, _ = b.displace_grid(0, 64)                          # synchronises
            b.synchronize()
            ref.compute_waves(times[-1])
            d, q = ref.read_maps()
            d2, q2 = b.read_maps()
            assert np.array_equal(d, d2) and np.array_equal(q, q2), frames
            assert np.array_equal(pinned[0], d[0]) and np.array_equal(pinned[1], q[0]), frames
            assert np.array_equal(staging[48:48 + n * n * 16].view(np.float32).reshape(n, n, 4), d[0])
            assert np.array_equal(staging[48 + n * n * 16:].view(np.float32).reshape(n, n, 4), q[0])
            rpos, _ = ref.displace_grid(0, 64)
            assert np.array_equal(pos, rpos), frames
            pd, pq = b.device_maps()                                 # zero-copy pointers of that same frame's maps
            assert pd and pq and pd != pq
    ref.close(); b.close()


@pytest.mark.parametrize("n,tiles", [(64, 2048), (256, 160), (512, 40)])
def test_streamed_variants_are_bit_identical_to_the_plain_ones(n, tiles):
    """Big batches at depth 2 run the streamed instantiations of all three kernels (non-temporal map stores, streamed
    intermediates); a synchronous call and a one-tile context run the plain ones.  Same frame, same bits: the kernels'
    arithmetic is written so that no instantiation may round differently (constant multiplies of the butterflies as
    instructions of their own, contraction off in the packing code) -- 256^2 in batches of 256 once differed by an ulp."""
    b = make_batch(n, tiles=tiles, depth=2)
    for j in range(3):
        b.compute_waves_async(0.3 * j)
    b.compute_waves_async(1.25)                      # streamed variants (8.4 M texels in flight twice: beyond the cache)
    b.synchronize()
    d2, q2 = b.read_maps()
    h2 = [b.heights(i) for i in (0, tiles - 1)]
    b.set_pipeline_depth(1)
    b.compute_waves(1.25)                            # plain intermediates, maps streamed or not by size
    d1, q1 = b.read_maps()
    assert np.array_equal(d1, d2) and np.array_equal(q1, q2)
    assert h2 == [b.heights(i) for i in (0, tiles - 1)]
    b.close()
    for i in (0, tiles // 2, tiles - 1):             # and a context of one tile (plain everything)
        s = make_batch(n, tiles=1, depth=1, seed=SEED + i)
        s.compute_waves(1.25)
        ds, qs = s.read_maps()
        assert np.array_equal(ds[0], d2[i]) and np.array_equal(qs[0], q2[i]), i
        s.close()
