"""Every kernel variant the frame launcher can select meets the oracle (through the C ABI).

One ComputeWaves (reference WSTessendorf.cpp:284-455) is three launches, and the host picks an instantiation of each per
frame -- store policy of the maps (NTS) and of the intermediates (ZNT), half2 intermediates (Z16), the Jacobian role (JAC),
one or two spectrum columns per z-pass workgroup (ZW), two- or single-transform batches (k_zpass / k_zpass_c1, round 4), and
two wave-uniform branches inside the z pass (fp16 copy of the spectrum, fp32 dispersion array) -- from the tile size, the
batch size, the pipeline depth, the mode and the precisions (ocean_api.hip: enqueue_frame, ocean_launch.h: launch_frame).  `ocean_last_launch` reports what a
frame really launched.  This file

  * drives, per tile size with a z-pass code path of its own (64: two-transform batches; 256, 512: four-transform batches;
    1024: two-transform batches for a lone tile, single-transform batches for a batch, the two-column form when streamed;
    2048: single-transform batches, the two-column form when streamed; 4096: always single-transform batches), every combination
    of store policy x intermediate precision x mode (FULL7 / JACOBIAN) x spectrum precision x dispersion width the
    launcher can select, checks tile 0 of each frame against the float64-FFT oracle, and asserts that the set of variants
    that met the oracle equals the set the launcher can select (a new variant without a test here fails);
  * asserts that the store policies of one configuration deliver the same bits;
  * names the regimes round 2 shipped untested: OCEAN_MODE_JACOBIAN at 4096^2 (serial and depth 2)
    and as an 8 x 1024^2 batch at depth 2 (streamed intermediates + two-column z pass), fp32 and half2; the fp32-dispersion
    fallback at 1024 and 2048.

Tolerances: per channel max|err| <= 1e-5 * max|channel| for the fp32 path (amplitude 1e-6); 1e-3 stated for the half2
intermediates and for the fp16 spectrum (tests/test_parity_gpu.py states the measured figures).
"""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL, TOL_AMP, TOL16 = 1e-5, 1e-6, 1e-3
SEED = 0x5EED0000
T_FRAME = 1.7
LONG_PERIOD = 4.0e5          # dispersion multiples beyond 16 bits at every size used here -> the fp32 array



def _compute_units():
    import torch
    return torch.cuda.get_device_properties(0).multi_processor_count


def handoff_grid_fits(n, one_launch=False):
    """The launcher gives a frame an in-launch hand-off (merged x pass, one-launch frame) only where every workgroup of that grid has a compute
    unit to itself on THIS device (ocean_launch.h, OceanTuning::handoff_wg_per_cu): the expectation follows the device's CU count, not the tile
    size alone.  x passes: C = 4 rows per workgroup below 4096."""
    c = 4 if n < 4096 else 2
    nu = n // 2 + 1
    hb, nb = (nu + 2 * c - 1) // (2 * c), (nu + c - 1) // c
    return (nu if one_launch else 0) + hb + 2 * nb <= _compute_units()


def chan_err(a, b):
    out = []
    for c in range(4):
        den = max(float(np.abs(b[..., c]).max()), 1e-30)
        out.append(float(np.abs(a[..., c].astype(np.float64) - b[..., c]).max()) / den)
    return out


class OracleCache:
    """One oracle per (n, period); one float64-FFT frame per (n, period, mode)."""

    def __init__(self):
        self.xi, self.oracles, self.frames = {}, {}, {}

    def get_xi(self, n):
        if n not in self.xi:
            import watersurfacerendering_amd as W
            b = W.OceanBatch(n, 1, 0)
            b.prepare(SEED)
            self.xi[n] = b.read_xi(0)
            b.close()
        return self.xi[n]

    def frame(self, n, period, jac):
        from oracle import oracle as O
        key = (n, period, jac)
        if key not in self.frames:
            if (n, period) not in self.oracles:
                o = O.Oracle(n, 1000.0, anim_period=period)
                o.prepare(xi=self.get_xi(n))
                self.oracles[(n, period)] = o
            o = self.oracles[(n, period)]
            a, d, q = o.compute_waves(T_FRAME, mode=O.MODE_JACOBIAN if jac else O.MODE_FULL7, fft=O.FFT_F64)
            self.frames[key] = (a, d.copy(), q.copy(), o.min_height, o.max_height)
        return self.frames[key]

    def drop(self, n):
        for d in (self.oracles, self.frames):
            for k in [k for k in d if k[0] == n]:
                del d[k]
        self.xi.pop(n, None)


@pytest.fixture(scope="module")
def oracles():
    return OracleCache()


def run_frame(n, tiles, depth, jac, inter_bits, h0_bits, period):
    """Tile 0's maps, heights and the launch records of one frame at T_FRAME in the given configuration."""
    import watersurfacerendering_amd as W
    from watersurfacerendering_amd import _abi
    b = W.OceanBatch(n, tiles, 0)
    b.set_params(anim_period=period)
    b.set_intermediate_precision(inter_bits)
    b.set_spectrum_precision(h0_bits)
    b.set_mode(_abi.OCEAN_MODE_JACOBIAN if jac else _abi.OCEAN_MODE_FULL7)
    b.set_pipeline_depth(depth)
    b.prepare(SEED)
    for j in range(depth - 1):                 # fill the other chains, so that the checked frame runs beside frames in flight
        b.compute_waves_async(0.3 * j)
    if depth > 1:
        b.compute_waves_async(T_FRAME)
        b.synchronize()
    else:
        b.compute_waves(T_FRAME)
    launches = b.last_launch()
    d, q = b.read_maps(0, 1)
    h = b.heights(0)
    b.close()
    return d[0], q[0], h, launches


def check_against_oracle(d, q, h, ref, jac, tol, what):
    ao, do, no, mn, mx = ref
    tol_amp = TOL_AMP if tol <= TOL else tol
    assert abs(h[0] - ao) <= tol_amp * abs(ao), (what, h[0], ao)
    assert abs(h[1] - mn) <= tol_amp * abs(ao) and abs(h[2] - mx) <= tol_amp * abs(ao), what
    assert np.all(np.isfinite(d)) and np.all(np.isfinite(q)), what
    ed, en = chan_err(d, do), chan_err(q, no)
    assert max(ed) <= tol, (what, "displacement", ed)
    assert max(en) <= tol, (what, "normal", en)
    if not jac:
        assert np.all(d[..., 3] == 1.0), what


def policies(n):
    """(name, tiles, depth) of the store policies the launcher can reach at tile size n (ocean_api.hip: enqueue_frame):
    plain   serial frame, maps below 200 MB                       -> plain map stores, plain intermediates
    nts     frames in flight (depth 2), little resident data      -> non-temporal maps, plain intermediates
            (4096^2: a serial frame -- its maps alone exceed the memory-side cache)
    stream  depth 2 and 16.7 M texels in flight                   -> non-temporal maps and intermediates"""
    big = max(1, (4096 * 4096) // (n * n))
    if n == 4096:
        return [("nts", 1, 1), ("stream", 1, 2)]
    if n == 1024:       # (a batch of 1024^2 tiles with plain stores takes the single-transform z pass, a lone tile the two-transform one)
        return [("plain", 1, 1), ("batch", 2, 1), ("nts", 1, 2), ("stream", big, 2)]
    return [("plain", 1, 1), ("nts", 1, 2), ("stream", big, 2)]


def expected_variants(n):
    from watersurfacerendering_amd import _abi as A
    out = set()
    for (name, tiles, _), z16, jac, h16, w32 in itertools.product(policies(n), (False, True), (False, True), (False, True), (False, True)):
        common = (A.OCEAN_LAUNCH_HALF_INTER if z16 else 0) | (A.OCEAN_LAUNCH_JACOBIAN if jac else 0)
        # ocean_kernels.h: zpass_c1_pays -- 4096^2 always; 2048^2 and batches of 1024^2 unless the intermediates are streamed
        c1 = n == 4096 or (name != "stream" and (n == 2048 or (n == 1024 and tiles >= 2)))
        # round 5: a SERIAL frame's fp32 intermediates go out write-through in the single-transform form at 1024 and 2048 (ocean_kernels.h: store_z)
        wt = c1 and name in ("plain", "batch") and n in (1024, 2048) and not z16
        zf = common | (A.OCEAN_LAUNCH_NT_INTER if name == "stream" else 0) | (A.OCEAN_LAUNCH_FP16_SPECTRUM if h16 else 0) | \
            (A.OCEAN_LAUNCH_FP32_DISPERSION if w32 else 0) | (A.OCEAN_LAUNCH_SINGLE_TRANSFORM if c1 else 0) | (A.OCEAN_LAUNCH_WT_INTER if wt else 0)
        xf = common | (A.OCEAN_LAUNCH_NT_MAPS if name not in ("plain", "batch") else 0)
        zw = 2 if n in (1024, 2048) and name == "stream" else 1
        out.add((n, zf, zw, xf, xf))
    return out


@pytest.mark.parametrize("n", [64, 256, 512, 1024, 2048, 4096])
def test_every_selectable_variant_meets_the_oracle(n, oracles):
    from watersurfacerendering_amd import _abi as A
    seen = set()
    split_seen = False
    for z16, jac, h16, w32 in itertools.product((False, True), (False, True), (False, True), (False, True)):
        period = LONG_PERIOD if w32 else 200.0
        ref = oracles.frame(n, period, jac)
        tol = TOL16 if (z16 or h16) else TOL
        same = None
        for name, tiles, depth in policies(n):
            what = (n, name, "half2" if z16 else "fp32", "jacobian" if jac else "full7", "fp16 spectrum" if h16 else "fp32 spectrum",
                    "fp32 dispersion" if w32 else "16-bit dispersion")
            d, q, h, launches = run_frame(n, tiles, depth, jac, 16 if z16 else 32, 16 if h16 else 32, period)
            check_against_oracle(d, q, h, ref, jac, tol, what)
            z, xb, xd = launches
            assert z["tile_size"] == n and z["grid_y"] == tiles and z["mode"] == (3 if jac else 0), what
            split_seen |= bool(z["flags"] & A.OCEAN_LAUNCH_SPLIT_LAST_ROUND)
            stag = A.OCEAN_LAUNCH_STAGGERED_START                  # not a variant: the same instantiation, started differently
            # nor is the merged x pass (round 5): k_xpass_b's instantiation with its DISP workgroups in the same launch -- single small tiles, not the Jacobian mode
            merged = A.OCEAN_LAUNCH_MERGED_X
            assert bool(xb["flags"] & merged) == bool(xd["flags"] & merged) == (n <= (128 if depth == 1 else 512) and tiles == 1 and not jac and handoff_grid_fits(n)), what
            # ... and the whole frame as ONE launch (k_frame: the bodies of k_zpass and k_xpass_b in one grid): pipelined frames of one tile up to
            # 128^2 in the usual form (fp32 spectrum, 16-bit dispersion, fp32 intermediates)
            one = A.OCEAN_LAUNCH_ONE_LAUNCH
            want_one = n <= 128 and tiles == 1 and depth > 1 and not (jac or z16 or h16 or w32) and handoff_grid_fits(n, one_launch=True)
            assert all(bool(li["flags"] & one) == want_one for li in (z, xb, xd)), what
            if want_one:        # (its launch record carries no store-policy flags of the z pass: mapped onto the three-launch variant it replaces)
                assert z["grid_x"] == xb["grid_x"] == xd["grid_x"] and xb["flags"] & A.OCEAN_LAUNCH_NT_MAPS, what
                z = dict(z, flags=z["flags"] & ~one)
                xb = dict(xb, flags=xb["flags"] & ~one); xd = dict(xd, flags=xd["flags"] & ~one)
            if xb["flags"] & merged:
                assert xb["grid_x"] == xd["grid_x"] and (xb["flags"] & ~A.OCEAN_LAUNCH_NT_MAPS) == (xd["flags"] & ~A.OCEAN_LAUNCH_NT_MAPS), what
            assert not any(li["flags"] & A.OCEAN_LAUNCH_SPLIT_ORDER for li in (z, xb, xd)), what      # (developer builds only: profiles/r05_4096_experiments.txt)
            for li in (z, xb, xd):
                assert bool(li["flags"] & A.OCEAN_LAUNCH_STAGGERED_START) == (n == 2048 and tiles == 1 and (li is not z or bool(z["flags"] & A.OCEAN_LAUNCH_SINGLE_TRANSFORM))), what
            seen.add((n, z["flags"] & ~(A.OCEAN_LAUNCH_SPLIT_LAST_ROUND | stag), z["per_workgroup"], xb["flags"] & ~(stag | merged), xd["flags"] & ~(stag | merged)))
            # the store policies (and with them the one- and two-column z pass, the split last round) never change a bit
            if same is None:
                same = (d, q, h)
            else:
                assert np.array_equal(same[0], d) and np.array_equal(same[1], q) and same[2] == h, what
    want = expected_variants(n)
    assert seen == want, {"never launched": sorted(want - seen), "launched but not expected": sorted(seen - want)}
    assert not split_seen                   # (rounds 2-3 split the last round of a serial 2048^2 z pass; the single-transform form replaced it)
    oracles.drop(n)


@pytest.mark.parametrize("bits", [32, 16])
@pytest.mark.parametrize("n,tiles,depth", [(4096, 1, 1), (4096, 1, 2), (1024, 8, 2)])
def test_jacobian_mode_in_the_streamed_regimes(n, tiles, depth, bits, oracles):
    """OCEAN_MODE_JACOBIAN (the intent of WSTessendorf.cpp:330-335, 421-428) where round 2 never checked it: 4096^2 serial and
    at depth 2 (two-column z pass, streamed intermediates), and BASELINE config 5's share -- 8 x 1024^2 per launch -- at depth
    2; every tile of the batch bit-identical to a one-tile, serial, plain-store context."""
    import watersurfacerendering_amd as W
    from watersurfacerendering_amd import _abi as A
    d, q, h, launches = run_frame(n, tiles, depth, True, bits, 32, 200.0)
    z, xb, xd = launches
    assert z["per_workgroup"] == (2 if n < 4096 and depth == 2 else 1)
    assert bool(z["flags"] & A.OCEAN_LAUNCH_SINGLE_TRANSFORM) == (n == 4096 or depth == 1)
    assert bool(z["flags"] & A.OCEAN_LAUNCH_NT_INTER) == (depth == 2)
    assert z["flags"] & A.OCEAN_LAUNCH_JACOBIAN and xb["flags"] & A.OCEAN_LAUNCH_JACOBIAN and xd["flags"] & A.OCEAN_LAUNCH_JACOBIAN
    check_against_oracle(d, q, h, oracles.frame(n, 200.0, True), True, TOL16 if bits == 16 else TOL, (n, tiles, depth, bits))
    assert float(np.abs(d[..., 3] - 1.0).max()) > 0.05                       # the Jacobian channel, not the constant 1
    # the whole batch against plain one-tile contexts, bit for bit
    b = W.OceanBatch(n, tiles, 0)
    b.set_intermediate_precision(bits); b.set_mode(A.OCEAN_MODE_JACOBIAN); b.set_pipeline_depth(depth); b.prepare(SEED)
    for j in range(depth):
        b.compute_waves_async(T_FRAME if j == depth - 1 else 0.3)
    b.synchronize()
    db, qb = b.read_maps()
    b.close()
    for i in sorted({0, tiles // 2, tiles - 1}):
        s = W.OceanBatch(n, 1, 0)
        s.set_intermediate_precision(bits); s.set_mode(A.OCEAN_MODE_JACOBIAN); s.prepare(SEED + i)
        s.compute_waves(T_FRAME)
        ds, qs = s.read_maps()
        if n < 4096:
            assert not (s.last_launch()[0]["flags"] & A.OCEAN_LAUNCH_NT_INTER) and s.last_launch()[0]["per_workgroup"] == 1
        s.close()
        assert np.array_equal(ds[0], db[i]) and np.array_equal(qs[0], qb[i]), i
    oracles.drop(n)


@pytest.mark.parametrize("n", [1024, 2048])
def test_fp32_dispersion_fallback_at_large_sizes(n, oracles):
    """A period so long that the dispersion's multiples of the base frequency exceed 16 bits (WSTessendorf.h:284-287 with
    T = 4e5 s): the z pass must read the fp32 array -- round 2 checked that branch at N = 64 only."""
    from watersurfacerendering_amd import _abi as A
    for jac in (False, True):
        d, q, h, launches = run_frame(n, 1, 1, jac, 32, 32, LONG_PERIOD)
        assert launches[0]["flags"] & A.OCEAN_LAUNCH_FP32_DISPERSION
        check_against_oracle(d, q, h, oracles.frame(n, LONG_PERIOD, jac), jac, TOL, (n, jac))
    oracles.drop(n)


def test_jacobian_buffers_are_allocated_by_the_first_frame_of_that_mode():
    """Contexts that never use OCEAN_MODE_JACOBIAN never pay for its three extra intermediates (+50 % of a chain)."""
    import torch
    import watersurfacerendering_amd as W
    from watersurfacerendering_amd import _abi as A
    n = 2048
    torch.cuda.synchronize()
    b = W.OceanBatch(n, 1, 0)
    b.prepare(SEED)
    b.compute_waves(0.5)
    free0, _ = torch.cuda.mem_get_info(0)
    b.compute_waves(1.0)
    free1, _ = torch.cuda.mem_get_info(0)
    assert free1 == free0                                  # steady state: frames allocate nothing
    b.set_mode(A.OCEAN_MODE_JACOBIAN)
    b.compute_waves(1.0)
    free2, _ = torch.cuda.mem_get_info(0)
    extra = free1 - free2
    want = (n // 2 + 1) * 2 * ((n // 2 + 16) & ~15) * 8 + 2 * ((n // 2 + 16) & ~15) * n * 4      # z3 + jraw + jac0
    assert want <= extra <= want + (32 << 20), (extra, want)    # allocation granularity
    b.compute_waves(1.5)
    assert torch.cuda.mem_get_info(0)[0] == free2
    b.close()



@pytest.mark.parametrize("n", [16, 64, 128, 256, 512])
def test_merged_x_pass_delivers_the_bits_of_the_three_launch_frame(n, oracles):
    """Round 5: frames of ONE small tile run the whole x axis as one launch -- k_xpass_b with its DISP workgroups, which
    transform pair 0 at once and wait for the tile's HEIGHT workgroups (raw heights handed over write-through inside the launch) only before
    their stores -- two launches per frame instead of three.  Against the three-launch frame (ocean_set_merged_xpass(ctx, 0)), bit for bit:
    every mode that can merge x fp32 / half2 intermediates x synchronous calls (tracked: the merged launch's last workgroup writes the
    records) / asynchronous frames at depth 1 and 4 (other chains' workgroups on the same compute units); the Jacobian mode and batches keep
    three launches; and the first configuration meets the oracle directly."""
    import watersurfacerendering_amd as W
    from watersurfacerendering_amd import _abi as A

    def frames(merged, mode, bits, depth, sync, tiles=1):
        b = W.OceanBatch(n, tiles, 0)
        b.set_intermediate_precision(bits); b.set_mode(mode); b.set_pipeline_depth(depth); b.set_merged_xpass(merged)
        b.prepare(SEED)
        amps = None
        if sync:
            for t in (0.4, 0.9, T_FRAME):
                amps = b.compute_waves(t)
        else:
            for j in range(2 * depth + 1):
                b.compute_waves_async(T_FRAME if j == 2 * depth else 0.3 * j)
            b.synchronize()
        launches = b.last_launch()
        d, q = b.read_maps()
        h = [b.heights(i) for i in range(tiles)]
        b.close()
        return d, q, h, amps, launches

    first = True
    for mode, bits, (depth, sync) in itertools.product((A.OCEAN_MODE_FULL7, A.OCEAN_MODE_CHOPPY5, A.OCEAN_MODE_HEIGHT1), (32, 16), ((1, True), (1, False), (4, False))):
        what = (n, mode, bits, depth, sync)
        d1, q1, h1, a1, l1 = frames(False, mode, bits, depth, sync)
        d2, q2, h2, a2, l2 = frames(True, mode, bits, depth, sync)
        assert not any(li["flags"] & A.OCEAN_LAUNCH_MERGED_X for li in l1), what
        # pipelined frames in the usual form go one step further: the whole frame as ONE launch (k_frame)
        assert all(bool(li["flags"] & A.OCEAN_LAUNCH_ONE_LAUNCH) == (n <= 128 and depth > 1 and bits == 32 and handoff_grid_fits(n, one_launch=True)) for li in l2), what
        if n > 128 and depth == 1:          # serial frames from 256^2 up keep three launches (the hand-off costs more than the boundary it replaces)
            assert not any(li["flags"] & A.OCEAN_LAUNCH_MERGED_X for li in l2), what
        else:
            assert l2[1]["flags"] & A.OCEAN_LAUNCH_MERGED_X and l2[2]["flags"] & A.OCEAN_LAUNCH_MERGED_X and l2[1]["grid_x"] > l1[1]["grid_x"], what
        assert np.array_equal(d1, d2) and np.array_equal(q1, q2) and h1 == h2, what
        if sync:
            assert np.array_equal(a1, a2) and h2[0][0] == float(a2[0]), what
        if first:
            check_against_oracle(d2[0], q2[0], h2[0], oracles.frame(n, 200.0, False), False, TOL, what)
            first = False
    # where it does not apply the frame keeps three launches (and says so): the Jacobian mode, a batch
    for mode, tiles in ((A.OCEAN_MODE_JACOBIAN, 1), (A.OCEAN_MODE_FULL7, 3)):
        _, _, _, _, l = frames(True, mode, 32, 1, True, tiles)
        assert not any(li["flags"] & A.OCEAN_LAUNCH_MERGED_X for li in l), (n, mode, tiles)
    oracles.drop(n)


def test_merged_x_pass_under_load_over_many_frames():
    """The hand-off inside the merged launch (write-through stores, one counted arrival per HEIGHT workgroup, a polled wait) under the
    conditions that expose a stale read: thousands of frames back to back at depth 4 -- other chains' workgroups on the same compute units,
    warm L1s -- every frame's amplitude and a checksum of both maps against the three-launch context fed the same times."""
    import watersurfacerendering_amd as W
    n, frames = 512, 1500
    ctx = []
    for merged in (False, True):
        b = W.OceanBatch(n, 1, 0)
        b.set_pipeline_depth(4); b.set_merged_xpass(merged); b.set_frame_tracking(True)
        b.prepare(SEED + 9)
        ctx.append(b)
    rng = np.random.default_rng(5)
    times = rng.uniform(0.0, 500.0, size=frames).astype(np.float32)
    for start in range(0, frames, 10):                 # ten frames in flight over four chains, then the last one's maps and heights
        out = []
        for b in ctx:
            for t in times[start:start + 10]:
                b.compute_waves_async(float(t))
            amp = b.wait_frame()[0]                    # (tracked: the merged launch's last workgroup wrote the record)
            d, q = b.read_maps()
            out.append((float(amp), b.heights(0), d.copy(), q.copy()))
        assert out[0][0] == out[1][0] and out[0][1] == out[1][1], start
        assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][3], out[1][3]), start
    for b in ctx:
        b.close()
