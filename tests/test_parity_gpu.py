"""GPU parity tests: the HIP path (called through the C ABI of include/ocean.h)
against the CPU oracle on identical (xi, params, t).

Tolerance (north_star: "within a stated float tolerance"):
    per output channel   max|err| <= 1e-5 * max|channel|
against the oracle run with float64 FFTs; amplitude A, min, max: relative 1e-6
of A.  The HIP path is fp32 throughout (observed error ~3e-7).
"""
import numpy as np
from conftest import pinned_array
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-5
TOL_AMP = 1e-6

ALT = dict(length=250.0, wind=(1.0, 0.0), wind_speed=10.0, lam=-2.0)


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


def make_gpu(n, xi, seed=0, length=1000.0, wind=(1.0, 1.0), wind_speed=30.0, anim_period=200.0,
             phillips_a=3e-7, damping=0.1, lam=-1.0, tiles=1, dispersion=None):
    import watersurfacerendering_amd as W
    b = W.OceanBatch(n, tiles, 0)
    if dispersion is not None:
        b.set_dispersion(*dispersion)
    b.set_params(tile_length=length, wind_dir_x=wind[0], wind_dir_y=wind[1], wind_speed=wind_speed,
                 anim_period=anim_period, phillips_const=phillips_a, damping=damping, lambda_=lam)
    b.prepare(seed, xi)
    return b


def check_frame(b, o, t, tile=0):
    from oracle import oracle as O
    ao, do, no = o.compute_waves(t, fft=O.FFT_F64)
    ag = float(b.compute_waves(t)[tile])
    dg, ng = b.read_maps(tile, 1)
    a, mn, mx = b.heights(tile)
    assert abs(ag - ao) <= TOL_AMP * abs(ao), (ag, ao)
    assert a == ag
    assert abs(mn - o.min_height) <= TOL_AMP * abs(ao) and abs(mx - o.max_height) <= TOL_AMP * abs(ao)
    ed, en = chan_err(dg[0], do), chan_err(ng[0], no)
    assert max(ed) <= TOL, ("displacement", ed)
    assert max(en) <= TOL, ("normal", en)
    assert np.all(dg[0][..., 3] == 1.0)
    return ed, en


@pytest.mark.parametrize("n", [16, 32, 64, 128, 256, 512, 1024])
def test_maps_match_oracle_defaults(n):
    from oracle import oracle as O
    xi = O.gauss_xi_numpy(1234, n)
    o = make_oracle(n, xi)
    b = make_gpu(n, xi[None])
    for t in (0.0, 1.5, 7.25, 1000.0):
        check_frame(b, o, t)
    b.close()


@pytest.mark.parametrize("n", [64, 512])
def test_negative_and_very_large_times(n):
    """omega*t beyond the fast range reduction (|x| >= 1e5 rad takes the library sincos path), and t < 0."""
    from oracle import oracle as O
    xi = O.gauss_xi_numpy(4321, n)
    o = make_oracle(n, xi)
    b = make_gpu(n, xi[None])
    for t in (-3.25, 9.0e3, 5.0e4, 3.0e5, 2.5e6):
        check_frame(b, o, t)
    b.close()


@pytest.mark.parametrize("dispersion", [(1, 2.0), (1, 40.0), (2, 5.0)])
@pytest.mark.parametrize("n", [64, 256])
def test_other_dispersion_relations(n, dispersion):
    """the finite-depth and small-wave relations the reference defines (WSTessendorf.h:301-315)
    but never calls.  Device omega against the oracle's (the finite-depth one goes through a double tanh on both
    sides: at most a couple of quantisation steps may differ), then the maps."""
    from oracle import oracle as O
    xi = O.gauss_xi_numpy(2468, n)
    o = make_oracle(n, xi, dispersion=dispersion)
    b = make_gpu(n, xi[None], dispersion=dispersion)
    _, om = b.read_spectrum(0)
    differing = int((om != o.omega).sum())
    assert differing <= (2 if dispersion[0] == 1 else 0), differing
    deep = make_oracle(n, xi)
    assert not np.array_equal(deep.omega, o.omega)                  # the option has an effect at these parameters
    if differing == 0:
        for t in (0.0, 2.5, 400.0):
            check_frame(b, o, t)
    import watersurfacerendering_amd as W
    with pytest.raises(W.OceanError):
        b.set_dispersion(3, 1.0)
    with pytest.raises(W.OceanError):
        b.set_dispersion(1, 0.0)
    b.close()


def test_huge_animation_period_uses_fp32_dispersion():
    """The frame kernels read omega as a 16-bit multiple of the base frequency; with a period so long that the
    multiples exceed 16 bits (here ~1e5 steps) the context must fall back to the fp32 array."""
    from oracle import oracle as O
    n = 64
    xi = O.gauss_xi_numpy(777, n)
    o = make_oracle(n, xi, anim_period=4.0e5)
    assert float(o.omega.max()) / (2 * np.pi / 4.0e5) > 65536
    b = make_gpu(n, xi[None], anim_period=4.0e5)
    _, om = b.read_spectrum(0)
    assert np.array_equal(om, o.omega)
    for t in (0.0, 1.5, 300.0):
        check_frame(b, o, t)
    b.close()


@pytest.mark.parametrize("n", [16, 64, 256, 512])
def test_maps_match_oracle_alt_params(n):
    from oracle import oracle as O
    xi = O.gauss_xi_numpy(99, n)
    o = make_oracle(n, xi, **ALT)
    b = make_gpu(n, xi[None], **ALT)
    for t in (0.0, 3.5, 250.0):
        check_frame(b, o, t)
    b.close()


@pytest.mark.parametrize("n", [64, 512])
def test_device_init_matches_oracle(n):
    """Prepare() on the device: generated draws, h0 and omega (1e-6 rel; omega exact)."""
    from oracle import oracle as O
    o = O.Oracle(n)
    o.prepare(seed=0x5EED0000)
    b = make_gpu(n, None, seed=0x5EED0000)
    xi = b.read_xi(0)
    assert np.abs(xi - o.xi).max() <= 1e-6 * np.abs(o.xi).max()
    h0, om = b.read_spectrum(0)
    assert np.array_equal(om, o.omega), float(np.abs(om - o.omega).max())
    assert np.abs(h0 - o.h0).max() <= 2e-6 * np.abs(o.h0).max()
    b.close()


# ---------------------------------------------------------------------------
# committed golden fixtures (inputs + expected outputs; tests/golden/make_golden.py)
import glob
import os
import subprocess

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "ocean_n*.npz"))))
def test_golden_fixtures_on_gpu(path):
    g = np.load(path)
    n = int(g["n"])
    b = make_gpu(n, g["xi"][None], length=float(g["length"]), wind=tuple(map(float, g["wind"])),
                 wind_speed=float(g["wind_speed"]), lam=float(g["lam"]))
    h0, om = b.read_spectrum(0)
    assert np.array_equal(om, g["omega"])
    assert np.abs(h0 - g["h0"]).max() <= 2e-6 * np.abs(g["h0"]).max()
    for i, t in enumerate(g["times"]):
        amp = float(b.compute_waves(float(t))[0])
        d, q = b.read_maps()
        assert abs(amp - float(g[f"amp{i}"])) <= TOL_AMP * amp * 2
        assert max(chan_err(d[0], g[f"disp{i}"])) <= TOL
        assert max(chan_err(q[0], g[f"nrm{i}"])) <= TOL
    b.close()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "sampled_n*.npz"))))
def test_sampled_golden_fixtures_on_gpu(path):
    """SURVEY.md 8c, N = 256, 512, 1024: amplitude, per-channel mean |value| and 1024 LCG-sampled texels of the
    committed fixtures, with the draws generated ON THE DEVICE from the seed (the fixture holds no xi)."""
    g = np.load(path)
    n = int(g["n"])
    b = make_gpu(n, None, seed=int(g["seed"]))
    idx = g["index"]
    for i, t in enumerate(g["times"]):
        amp = float(b.compute_waves(float(t))[0])
        d, q = b.read_maps()
        d, q = d[0].reshape(-1, 4), q[0].reshape(-1, 4)
        scale = np.maximum(g[f"maxabs{i}"], 1e-30)
        assert abs(amp - float(g[f"amp{i}"])) <= 2 * TOL_AMP * amp
        mean = np.concatenate([np.abs(d).mean(0, dtype=np.float64), np.abs(q).mean(0, dtype=np.float64)])
        assert np.all(np.abs(mean - g[f"meanabs{i}"]) <= TOL * scale)
        assert np.all(np.abs(d[idx] - g[f"disp{i}"]) <= TOL * scale[:4])
        assert np.all(np.abs(q[idx] - g[f"nrm{i}"]) <= TOL * scale[4:])
    b.close()


# ---------------------------------------------------------------------------
# full BASELINE sizes: direct oracle comparison + size-independent properties
@pytest.mark.parametrize("n", [2048, 4096])
def test_full_size_against_oracle_and_properties(n):
    from oracle import oracle as O
    seed = 0x5EED0000
    b = make_gpu(n, None, seed=seed)
    xi = b.read_xi(0)
    h0, om = b.read_spectrum(0)
    o = make_oracle(n, xi)
    assert np.array_equal(om, o.omega)
    t = 2.75
    ed, en = check_frame(b, o, t)
    d, q = b.read_maps()
    d, q = d[0].astype(np.float64), q[0].astype(np.float64)
    amp = b.heights(0)[0]
    # (1) zero DC amplitude => every field has zero spatial mean
    for a in (d[..., 0], d[..., 1], d[..., 2], q[..., 0], q[..., 1], q[..., 2], q[..., 3]):
        assert abs(a.mean()) <= 2e-6 * np.abs(a).max()
    # (2) Parseval for the height: sum h^2 = N^2 * sum |H_h|^2, H_h = (H(k) + H(-k))/2
    wt = (om * np.float32(t)).astype(np.float32).astype(np.float64)
    H = 2.0 * (h0[..., 0].astype(np.float64) * np.cos(wt) - h0[..., 1].astype(np.float64) * np.sin(wt))
    Hm = np.roll(H[::-1, ::-1], (1, 1), axis=(0, 1))          # H at ((N-m)%N, (N-n)%N)
    Hh = 0.5 * (H + Hm)
    lhs = float(np.sum((d[..., 1] * amp) ** 2))
    rhs = float(n * n * np.sum(Hh ** 2))
    assert abs(lhs - rhs) <= 1e-5 * rhs
    # (3) sampled texels against the defining DFT sum in float64 (O(N^2) each)
    k1 = o.kvec[0, :, 0].astype(np.float64)
    idx = np.arange(n)
    rng = np.random.default_rng(n)
    for p, qq in rng.integers(0, n, size=(3, 2)):
        ph = np.exp(2j * np.pi * (p * idx[:, None] + qq * idx[None, :]) / n)
        s = -1.0 if (p + qq) & 1 else 1.0
        hval = s * float(np.sum(H * ph).real)
        sxval = s * float(np.sum(1j * k1[None, :] * H * ph).real)
        assert abs(hval / amp - d[p, qq, 1]) <= TOL * 1.0
        assert abs(sxval - q[p, qq, 0]) <= TOL * np.abs(q[..., 0]).max()
    b.close()


def test_full_size_animation_loop_and_linearity_in_xi():
    """Size-independent properties at BASELINE's headline size (tests/test_oracle.py states and derives them for the oracle):
    the animation loops with the period T -- frame(t + T) = frame(t) to 1e-4 of a channel's maximum, the rounding of the single fp32
    product omega * t (.h:267) -- and every raw field is linear in xi: doubling xi doubles displacements, normals and A (1e-6: a power of two
    commutes with fp32 rounding) and leaves the normalised height unchanged."""
    n = 2048
    period = 200.0
    b = make_gpu(n, None, seed=0x5EED0000)
    xi = b.read_xi(0)
    frames = {}
    for t in (0.0, 3.25, 3.25 + period, period, 0.5 * period):
        a = float(b.compute_waves(t)[0])
        d, q = b.read_maps()
        frames[t] = (a, d[0].copy(), q[0].copy())
    for t in (0.0, 3.25):
        (a0, d0, q0), (a1, d1, q1) = frames[t], frames[t + period]
        assert abs(a1 - a0) <= 1e-4 * a0
        for c in (0, 1, 2):
            assert np.abs(d1[..., c] - d0[..., c]).max() <= 1e-4 * np.abs(d0[..., c]).max(), ("displacement", c)
        for c in range(4):
            assert np.abs(q1[..., c] - q0[..., c]).max() <= 1e-4 * np.abs(q0[..., c]).max(), ("normal", c)
    assert np.abs(frames[0.5 * period][1][..., 0] - frames[0.0][1][..., 0]).max() > 0.1 * np.abs(frames[0.0][1][..., 0]).max()
    b2 = make_gpu(n, (2.0 * xi)[None])
    a2 = float(b2.compute_waves(3.25)[0])
    d2, q2 = b2.read_maps()
    a, d, q = frames[3.25]
    assert abs(a2 - 2.0 * a) <= 1e-6 * a2
    assert np.abs(q2[0] - 2.0 * q).max() <= 1e-6 * np.abs(q2[0]).max()
    assert np.abs(d2[0][..., [0, 2]] - 2.0 * d[..., [0, 2]]).max() <= 1e-6 * np.abs(d2[0][..., [0, 2]]).max()
    assert np.abs(d2[0][..., 1] - d[..., 1]).max() <= 1e-6
    b.close(); b2.close()


def test_lambda_scales_displacement_only():
    n = 256
    b = make_gpu(n, None, seed=5)
    b.compute_waves(1.0)
    d1, q1 = b.read_maps()
    b.set_lambda(-2.0)                      # SetLambda: no Prepare needed (.cpp:497-500)
    b.compute_waves(1.0)
    d2, q2 = b.read_maps()
    assert np.array_equal(q1, q2) and np.array_equal(d1[..., 1], d2[..., 1])
    assert np.allclose(d2[..., 0], 2.0 * d1[..., 0], rtol=1e-6, atol=0)
    assert np.allclose(d2[..., 2], 2.0 * d1[..., 2], rtol=1e-6, atol=0)
    b.close()


def test_zero_spectrum_minmax_quirk():
    """Reference quirk: max starts at FLT_MIN (.cpp:289-290) => A = FLT_MIN for a flat sea."""
    b = make_gpu(32, None, seed=1, phillips_a=0.0)
    amp = float(b.compute_waves(2.0)[0])
    a, mn, mx = b.heights(0)
    tiny = float(np.finfo(np.float32).tiny)
    assert amp == tiny and mx == tiny and mn == 0.0
    d, q = b.read_maps()
    assert np.all(d[..., 1] == 0.0) and np.all(d[..., 3] == 1.0) and np.all(q == 0.0)
    b.close()


def test_batch_tiles_are_independent_and_match_single_tile_runs():
    from oracle import oracle as O
    n, tiles, seed = 128, 5, 700
    b = make_gpu(n, None, seed=seed, tiles=tiles)
    offs = np.array([0.0, 0.5, 1.0, 10.0, 100.0], np.float32)
    b.set_time_offsets(offs)
    amps = b.compute_waves(1.0)
    d, q = b.read_maps()
    for i in range(tiles):
        xi = O.gauss_xi_numpy(seed + i, n)        # tile i uses seed + i
        assert np.abs(b.read_xi(i) - xi).max() <= 1e-6 * np.abs(xi).max()
        o = make_oracle(n, b.read_xi(i))
        ao, do, no = o.compute_waves(float(np.float32(1.0) + offs[i]), fft=O.FFT_F64)
        assert abs(float(amps[i]) - ao) <= TOL_AMP * ao
        assert max(chan_err(d[i], do)) <= TOL and max(chan_err(q[i], no)) <= TOL
    b.set_time_offsets(None)
    b.close()


def test_python_mirror_and_resize():
    """WSTessendorf mirror: reference call order, SetTileSize + Prepare, ignored bad size."""
    import watersurfacerendering_amd as W
    from oracle import oracle as O
    ws = W.WSTessendorf(64, 500.0)
    ws.SetWindDirection((0.0, 2.0)); ws.SetWindSpeed(12.0); ws.SetDamping(0.2); ws.SetPhillipsConst(5e-7)
    ws.SetAnimationPeriod(100.0); ws.SetLambda(-0.5)
    ws.SetTileSize(100)                       # not a power of two: ignored (.cpp:463-467)
    assert ws.GetTileSize() == 64
    assert ws.GetWindDir() == pytest.approx((0.0, 1.0))
    ws.SetTileSize(128)
    xi = O.gauss_xi_numpy(8, 128)
    ws.Prepare(seed=8, xi=xi[None])
    assert ws.GetDisplacementCount() == 128 * 128 and np.all(ws.GetNormals()[..., 1] == 1.0)
    amp = ws.ComputeWaves(4.0)
    o = make_oracle(128, xi, length=500.0, wind=(0.0, 2.0), wind_speed=12.0, damping=0.2, phillips_a=5e-7,
                    anim_period=100.0, lam=-0.5)
    ao, do, no = o.compute_waves(4.0, fft=O.FFT_F64)
    assert abs(amp - ao) <= TOL_AMP * ao
    assert max(chan_err(ws.GetDisplacements(), do)) <= TOL and max(chan_err(ws.GetNormals(), no)) <= TOL
    assert ws.GetMinHeight() == pytest.approx(o.min_height, rel=1e-5)
    assert ws.GetMaxHeight() == pytest.approx(o.max_height, rel=1e-5)


def test_errors_and_ordering():
    import watersurfacerendering_amd as W
    from watersurfacerendering_amd import _abi
    b = W.OceanBatch(64, 1, 0)
    with pytest.raises(W.OceanError) as e:
        b.compute_waves(0.0)                  # before Prepare
    assert e.value.code == _abi.OCEAN_E_NOT_READY
    with pytest.raises(W.OceanError):
        W.OceanBatch(64, 1, 99)               # no such device
    b.prepare(1)
    b.compute_waves(0.0)
    b.set_tile_size(32)                       # resize invalidates the prepared state
    with pytest.raises(W.OceanError):
        b.compute_waves(0.0)
    for bad in (lambda: b.set_mode(4), lambda: b.set_pipeline_depth(0), lambda: b.set_pipeline_depth(99),
                lambda: b.set_spectrum_precision(8)):
        with pytest.raises(W.OceanError):
            bad()
    assert b.kernel_names() == ["k_zpass", "k_xpass_b", "k_xpass_disp"]
    b.close()


def test_bind_output_and_async_stream_order():
    import torch
    import watersurfacerendering_amd as W
    n = 256
    b = W.OceanBatch(n, 1, 0)
    b.prepare(3)
    b.compute_waves(0.25)
    d_ref, q_ref = b.read_maps()
    maps = torch.zeros((2, n, n, 4), dtype=torch.float32, device="cuda:0")
    b.bind_output(maps[0].data_ptr(), maps[1].data_ptr())
    b.compute_waves_async(0.25)
    b.synchronize()
    got = maps.cpu().numpy()
    assert np.array_equal(got[0], d_ref[0]) and np.array_equal(got[1], q_ref[0])   # deterministic, bit for bit
    b.bind_output(None, None)
    b.close()


def test_frames_are_deterministic():
    b = make_gpu(512, None, seed=11)
    b.compute_waves(3.0); d1, q1 = b.read_maps()
    b.compute_waves(9.0)
    b.compute_waves(3.0); d2, q2 = b.read_maps()
    assert np.array_equal(d1, d2) and np.array_equal(q1, q2)
    b.close()


def test_cpp_adaptor_matches_python_binding(tmp_path):
    from watersurfacerendering_amd import _abi
    exe = tmp_path / "adaptor_demo"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "adaptor_demo.cpp"), "-o", str(exe),
                    "-L", os.path.dirname(_abi.LIB_PATH), "-locean_hip", "-Wl,-rpath," + os.path.dirname(_abi.LIB_PATH),
                    "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe), "128", "1.5", "0", "async"], capture_output=True, text=True, check=True)
    first, second = r.stdout.strip().splitlines()
    n, amp, mn, mx, sd, sn = first.split()
    tag, ok, us_amp, us_maps = second.split()       # ComputeWavesAsync() / Wait(): same frame as the blocking call, front pair untouched meanwhile
    assert tag == "async" and ok == "1", second
    # the same with the opt-in SelectFastestQueue() ahead of the first frame: not a character of the output may change
    r2 = subprocess.run([str(exe), "128", "1.5", "0", "noasync", "select"], capture_output=True, text=True, check=True)
    assert r2.stdout.strip().splitlines()[0] == first
    b = make_gpu(128, None, seed=42, wind=(1.0, 0.5), wind_speed=20.0, lam=-1.5)
    a = float(b.compute_waves(1.5)[0])
    d, q = b.read_maps()
    w7 = (np.arange(d[0].size) % 7 + 1).astype(np.float64)
    w5 = ((np.arange(q[0].size) + d[0].size) % 5 + 1).astype(np.float64)
    assert int(n) == 128 and float(amp) == pytest.approx(a, rel=1e-7)
    assert float(sd) == pytest.approx(float(np.sum(d[0].astype(np.float64).ravel() * w7)), rel=1e-9, abs=1e-6)
    assert float(sn) == pytest.approx(float(np.sum(q[0].astype(np.float64).ravel() * w5)), rel=1e-9, abs=1e-6)
    b.close()


# ---------------------------------------------------------------------------
# BASELINE config 4: fp16 spectrum vs fp32 spectrum, both against the float64-FFT oracle
FP16_TOL = 1e-3      # stated tolerance for the half2 spectrum: max|err| <= 1e-3 * max|channel| (measured ~2e-4)


@pytest.mark.parametrize("n", [512, 4096])
def test_fp16_spectrum_within_stated_tolerance(n):
    from oracle import oracle as O
    import watersurfacerendering_amd as W
    seed = 0x5EED0000
    b = W.OceanBatch(n, 1, 0)
    b.set_spectrum_precision(16)
    b.prepare(seed)
    xi = b.read_xi(0)
    o = make_oracle(n, xi)
    t = 4.5
    ao, do, no = o.compute_waves(t, fft=O.FFT_F64)
    a16 = float(b.compute_waves(t)[0])
    d16, n16 = b.read_maps()
    e16 = max(chan_err(d16[0], do)[:3] + chan_err(n16[0], no))
    assert abs(a16 - ao) <= FP16_TOL * ao
    assert e16 <= FP16_TOL, e16
    assert e16 > TOL / 10          # it really is the reduced-precision path
    b.set_spectrum_precision(32)
    with pytest.raises(W.OceanError):
        b.compute_waves(t)          # precision change needs Prepare(), like every spectrum parameter
    b.prepare(seed)
    a32 = float(b.compute_waves(t)[0])
    d32, n32 = b.read_maps()
    assert max(chan_err(d32[0], do) + chan_err(n32[0], no)) <= TOL
    assert abs(a32 - ao) <= TOL_AMP * ao
    b.close()


def test_pipeline_depth_two_gives_identical_frames():
    import watersurfacerendering_amd as W
    n = 512
    ref = W.OceanBatch(n, 1, 0); ref.prepare(21)
    pip = W.OceanBatch(n, 1, 0); pip.prepare(21); pip.set_pipeline_depth(2)
    for j in range(5):
        ref.compute_waves_async(0.3 * j)
        pip.compute_waves_async(0.3 * j)
    ref.synchronize(); pip.synchronize()
    d1, q1 = ref.read_maps(); d2, q2 = pip.read_maps()
    assert np.array_equal(d1, d2) and np.array_equal(q1, q2)
    assert ref.heights(0) == pip.heights(0)
    ref.close(); pip.close()


def test_pipelined_frames_keep_frame_order_semantics():
    """depth 4: frames run as independent chains; after synchronize the read-out refers to the
    frame enqueued last, and binding caller-owned output falls back to serial frames."""
    import torch
    import watersurfacerendering_amd as W
    n = 256
    ref = W.OceanBatch(n, 1, 0); ref.prepare(33)
    pip = W.OceanBatch(n, 1, 0); pip.prepare(33); pip.set_pipeline_depth(4)
    times = [0.1 * j for j in range(7)]
    for t in times:
        pip.compute_waves_async(t)
    pip.synchronize()
    ref.compute_waves(times[-1])
    d1, q1 = ref.read_maps(); d2, q2 = pip.read_maps()
    assert np.array_equal(d1, d2) and np.array_equal(q1, q2)
    assert ref.heights(0) == pip.heights(0)
    # synchronous call in pipelined mode still returns its own frame
    a = pip.compute_waves(0.55)
    b = ref.compute_waves(0.55)
    assert np.array_equal(a, b)
    # bound output => every frame lands in the caller's buffer, in order
    maps = torch.zeros((2, n, n, 4), dtype=torch.float32, device="cuda:0")
    pip.bind_output(maps[0].data_ptr(), maps[1].data_ptr())
    for t in times:
        pip.compute_waves_async(t)
    pip.synchronize()
    got = maps.cpu().numpy()
    assert np.array_equal(got[0], d1[0]) and np.array_equal(got[1], q1[0])
    pip.bind_output(None, None)
    ref.close(); pip.close()


def test_async_readout_into_registered_host_memory():
    """maps DMA'd straight into a pinned caller buffer laid out like the reference's
    staging buffer [displacements | normals] (WaterSurfaceMesh.cpp:705-744), ordered after the frame."""
    import watersurfacerendering_amd as W
    n = 512
    b = W.OceanBatch(n, 1, 0); b.prepare(17)
    staging = pinned_array((2, n, n, 4), np.float32)
    for t in (0.5, 1.0):
        b.compute_waves_async(t)
        b.read_maps_async(staging[0:1], staging[1:2])
        b.synchronize()
        d, q = b.read_maps()
        assert np.array_equal(staging[0], d[0]) and np.array_equal(staging[1], q[0])
    b.close()


@pytest.mark.parametrize("n", [256, 512, 4096])
def test_reduced_modes_match_oracle_modes(n):
    """BASELINE configs 1-2: HEIGHT1 (1 iFFT) and CHOPPY5 (5 iFFTs) against the oracle's same modes."""
    from oracle import oracle as O
    from watersurfacerendering_amd import _abi
    b = make_gpu(n, None, seed=404)
    o = make_oracle(n, b.read_xi(0))
    t = 2.25
    for gmode, omode in ((_abi.OCEAN_MODE_CHOPPY5, O.MODE_CHOPPY5), (_abi.OCEAN_MODE_HEIGHT1, O.MODE_HEIGHT1),
                         (_abi.OCEAN_MODE_FULL7, O.MODE_FULL7)):
        b.set_mode(gmode)
        ag = float(b.compute_waves(t)[0])
        dg, ng = b.read_maps()
        ao, do, no = o.compute_waves(t, mode=omode, fft=O.FFT_F64)
        assert abs(ag - ao) <= TOL_AMP * ao
        for c in range(4):
            for got, ref in ((dg[0][..., c], do[..., c]), (ng[0][..., c], no[..., c])):
                m = float(np.abs(ref).max())
                if m == 0.0:
                    assert np.all(got == 0.0)                    # fields the mode drops are exactly zero
                else:
                    assert float(np.abs(got - ref).max()) <= TOL * m
    b.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_call_sequences_match_oracle(seed):
    """Random interleavings of setters, Prepare, synchronous/asynchronous frames, pipeline depth,
    modes and output binding: after every synchronisation point the maps equal the oracle's for the
    state the reference object would be in."""
    import torch
    from oracle import oracle as O
    import watersurfacerendering_amd as W
    rng = np.random.default_rng(seed)
    n = int(rng.choice([32, 64, 128]))
    state = dict(length=1000.0, wind=(1.0, 1.0), wind_speed=30.0, anim_period=200.0, phillips_a=3e-7, damping=0.1, lam=-1.0)
    b = W.OceanBatch(n, 1, 0)
    xi = O.gauss_xi_numpy(seed, n)
    prepared = dict(state)
    b.prepare(seed, xi[None])
    mode = 0
    bound = None
    last_t = None
    log = []
    for step in range(150):
        op = rng.integers(0, 9)
        log.append(int(op))
        if op == 0:
            state["lam"] = float(rng.uniform(-2.5, -0.2)); b.set_lambda(state["lam"])
            last_t = None                     # takes effect at the next frame
        elif op == 1:
            state["wind_speed"] = float(rng.uniform(5, 40)); state["wind"] = (float(rng.uniform(-1, 1)), float(rng.uniform(0.1, 1)))
            state["damping"] = float(rng.uniform(0.05, 0.5))
            b.set_params(wind_speed=state["wind_speed"], wind_dir_x=state["wind"][0], wind_dir_y=state["wind"][1],
                         damping=state["damping"], lambda_=state["lam"])
            last_t = None
        elif op == 2:
            xi = O.gauss_xi_numpy(seed * 100 + step, n)
            b.prepare(0, xi[None]); prepared = dict(state)
            last_t = None                     # no frame of the new spectrum yet: nothing to compare
        elif op == 3:
            b.set_pipeline_depth(int(rng.integers(1, 5)))
        elif op == 4:
            mode = int(rng.integers(0, 3)); b.set_mode(mode)
            last_t = None                     # takes effect at the next frame
        elif op == 5:
            if bound is None:
                bound = torch.zeros((2, n, n, 4), dtype=torch.float32, device="cuda:0")
                b.bind_output(bound[0].data_ptr(), bound[1].data_ptr())
            else:
                b.bind_output(None, None); bound = None
            last_t = None                     # the other buffer holds an older frame
        elif op in (6, 7):
            for _ in range(int(rng.integers(1, 6))):
                last_t = float(rng.uniform(0, 50)); b.compute_waves_async(last_t)
        else:
            last_t = float(rng.uniform(0, 50)); b.compute_waves(last_t)
        if last_t is None or rng.random() < 0.5:
            continue
        b.synchronize()
        # reference state: spectrum parameters as of the last Prepare, lambda and mode as of now
        p = dict(prepared); p["lam"] = state["lam"]
        o = make_oracle(n, xi, **p)
        ao, do, no = o.compute_waves(last_t, mode=mode, fft=O.FFT_F64)
        dg, ng = b.read_maps()
        if bound is not None:
            got = bound.cpu().numpy()
            assert np.array_equal(got[0], dg[0]) and np.array_equal(got[1], ng[0])
        a, mn, mx = b.heights(0)
        assert abs(a - ao) <= TOL_AMP * ao * 2, (log, n, mode, last_t)
        for c in range(4):
            for got, ref in ((dg[0][..., c], do[..., c]), (ng[0][..., c], no[..., c])):
                m = float(np.abs(ref).max())
                assert float(np.abs(got - ref).max()) <= TOL * max(m, 1e-30) or (m == 0.0 and np.all(got == 0.0)), (log, c)
    b.close()


# ---------------------------------------------------------------------------
# BASELINE config 4's reduced mode: half2 intermediates between the two passes
Z16_TOL = 1e-3       # stated tolerance: max|err| <= 1e-3 * max|channel| against the float64-FFT oracle (measured ~2e-4)


@pytest.mark.parametrize("n,params", [(64, {}), (512, {}), (512, ALT), (2048, {}), (4096, {})])
def test_fp16_intermediates_within_stated_tolerance(n, params):
    """ocean_set_intermediate_precision(16): the z-pass outputs travel as scaled half2.  Within 1e-3 of every
    channel's maximum at several times (the scale is time independent: nothing may overflow at any t), really
    reduced precision, and the fp32 mode is untouched by the switch."""
    from oracle import oracle as O
    import watersurfacerendering_amd as W
    seed = 0x5EED0000
    kw = dict(params)
    b = make_gpu(n, None, seed=seed, **kw)
    xi = b.read_xi(0)
    o = make_oracle(n, xi, **kw)
    b.set_intermediate_precision(16)
    with pytest.raises(W.OceanError):
        b.compute_waves(0.0)            # needs Prepare(), like every spectrum parameter
    b.prepare(seed)
    assert b.algorithmic_bytes_per_texel == 59
    worst = 0.0
    for t in ((0.0, 4.5, 1000.0) if n <= 2048 else (4.5,)):
        ao, do, no = o.compute_waves(t, fft=O.FFT_F64)
        a16 = float(b.compute_waves(t)[0])
        d16, n16 = b.read_maps()
        assert np.all(np.isfinite(d16)) and np.all(np.isfinite(n16))
        e16 = max(chan_err(d16[0], do)[:3] + chan_err(n16[0], no))
        assert abs(a16 - ao) <= Z16_TOL * ao
        assert e16 <= Z16_TOL, (t, e16)
        worst = max(worst, e16)
    assert worst > TOL / 10             # it really is the reduced-precision path
    # pipelined frames in that mode equal serial ones bit for bit
    b.compute_waves(2.0); d1, q1 = b.read_maps()
    b.set_pipeline_depth(2)
    for t in (0.5, 1.0, 2.0):
        b.compute_waves_async(t)
    b.synchronize(); d2, q2 = b.read_maps()
    assert np.array_equal(d1, d2) and np.array_equal(q1, q2)
    b.set_pipeline_depth(1)
    b.set_intermediate_precision(32)
    b.prepare(seed)
    assert b.algorithmic_bytes_per_texel == 73
    check_frame(b, o, 4.5)
    b.close()


# ---------------------------------------------------------------------------
# Jacobian / foam channel
@pytest.mark.parametrize("n,lam", [(16, -1.0), (64, -1.7), (256, -1.0), (512, -2.0), (2048, -1.0)])
def test_jacobian_mode_matches_oracle(n, lam):
    """OCEAN_MODE_JACOBIAN: displacement.w = (1 + l dxDx)(1 + l dzDz) - (l dxDz)(l dzDx) (the intent of
    WSTessendorf.cpp:330-335, 421-428) against the oracle's two-transform restatement, within 1e-5 of max|w|;
    the seven reference fields are exactly what FULL7 gives; lambda reaches the Jacobian without Prepare; the mode
    combines with pipelining (bit-identical) and with the half2 intermediates (1e-3)."""
    from oracle import oracle as O
    from watersurfacerendering_amd import _abi
    seed = 0x5EED0000 + 5
    b = make_gpu(n, None, seed=seed, lam=lam)
    o = make_oracle(n, b.read_xi(0), lam=lam)
    t = 2.25
    b.compute_waves(t)
    d7, q7 = b.read_maps()
    b.set_mode(_abi.OCEAN_MODE_JACOBIAN)
    assert b.algorithmic_bytes_per_texel == 85
    ag = float(b.compute_waves(t)[0])
    dg, ng = b.read_maps()
    ao, do, no = o.compute_waves(t, mode=O.MODE_JACOBIAN, fft=O.FFT_F64)
    assert abs(ag - ao) <= TOL_AMP * ao
    assert max(chan_err(dg[0], do)) <= TOL and max(chan_err(ng[0], no)) <= TOL
    assert float(np.abs(do[..., 3] - 1.0).max()) > 0.05                      # the channel is not the constant 1 any more
    assert np.array_equal(ng, q7) and np.array_equal(dg[..., 0], d7[..., 0]) and np.array_equal(dg[..., 2], d7[..., 2])
    assert np.abs(dg[..., 1] - d7[..., 1]).max() <= 2e-6                     # the height goes through pair 3 instead of the real-row transform
    # SetLambda without Prepare (.cpp:497-500) reaches the Jacobian too
    b.set_lambda(0.5 * lam); o.set_lambda(0.5 * lam)
    b.compute_waves(t)
    d2, _ = b.read_maps()
    _, do2, _ = o.compute_waves(t, mode=O.MODE_JACOBIAN, fft=O.FFT_F64)
    assert max(chan_err(d2[0], do2)) <= TOL
    # pipelined: bit-identical
    b.set_pipeline_depth(3)
    for tt in (0.3, 0.6, t):
        b.compute_waves_async(tt)
    b.synchronize()
    d3, _ = b.read_maps()
    assert np.array_equal(d2, d3)
    b.set_pipeline_depth(1)
    # with the reduced-precision intermediates
    b.set_intermediate_precision(16)
    b.prepare(seed)
    assert b.algorithmic_bytes_per_texel == 69
    b.compute_waves(t)
    d16, n16 = b.read_maps()
    assert max(chan_err(d16[0], do2)) <= Z16_TOL and max(chan_err(n16[0], no)) <= Z16_TOL
    b.close()


def test_jacobian_reaches_the_vertex_stage_consumer():
    """displacement.w travels to the consumer like in the reference's shaders (WaterSurfaceMesh.vert:29): the
    vertex-stage kernel hands out the sampled Jacobian and the foam test of .frag:210-212 can be applied to it."""
    from oracle import consumer as C
    from watersurfacerendering_amd import _abi
    n, lam = 256, -2.5
    b = make_gpu(n, None, seed=77, lam=lam)
    b.set_mode(_abi.OCEAN_MODE_JACOBIAN)
    amp = float(b.compute_waves(1.0)[0])
    disp, nrm = b.read_maps()
    pos, nr = b.displace_grid(0, n, 1000.0 / 512.0, 1.0, lam)
    opos, onr = C.displace_grid(disp[0], nrm[0], amp, n, 1000.0 / 512.0, 1.0, lam)
    assert np.abs(pos - opos).max() <= 1e-6 * np.abs(opos).max()
    foam, ofoam = C.foam_mask(pos), C.foam_mask(opos)
    assert 0.0 < ofoam.mean() < 0.5
    assert (foam != ofoam).mean() <= 1e-4            # only vertices whose w rounds differently around 0 may differ
    b.close()


@pytest.mark.parametrize("n,seed", [(64, 123), (64, 7), (256, 99), (1024, 4242)])
def test_jacobian_with_half2_intermediates_cannot_overflow(n, seed):
    """Regression: pair 3 packs the height (unit weight) with the k-weighted cross derivative, and after ONE axis both
    parts of the transform hold a mixture of the two -- scaling the imaginary part like a k-weighted field overflowed
    the half2 form for some spectra (NaN maps).  Both parts now share one scale and the cross part is pre-amplified to
    the height's magnitude: finite and within the half2 mode's 1e-3 at seeds that used to fail."""
    from oracle import oracle as O
    from watersurfacerendering_amd import _abi
    for params in ({}, ALT):
        b = make_gpu(n, None, seed=seed, **params)
        o = make_oracle(n, b.read_xi(0), **params)
        b.set_intermediate_precision(16); b.set_mode(_abi.OCEAN_MODE_JACOBIAN); b.prepare(seed)
        for t in (0.0, 1.7, 300.0):
            ag = float(b.compute_waves(t)[0])
            d, q = b.read_maps()
            assert np.all(np.isfinite(d)) and np.all(np.isfinite(q)), (n, seed, t)
            ao, do, no = o.compute_waves(t, mode=O.MODE_JACOBIAN, fft=O.FFT_F64)
            assert abs(ag - ao) <= Z16_TOL * ao
            assert max(chan_err(d[0], do)) <= Z16_TOL and max(chan_err(q[0], no)) <= Z16_TOL, (n, seed, t, chan_err(d[0], do))
        b.close()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "jacobian_n*.npz"))))
def test_jacobian_golden_fixtures_on_gpu(path):
    g = np.load(path)
    base = np.load(os.path.join(GOLDEN, str(g["inputs"])))
    from watersurfacerendering_amd import _abi
    n = int(g["n"])
    b = make_gpu(n, base["xi"][None], length=float(base["length"]), wind=tuple(map(float, base["wind"])),
                 wind_speed=float(base["wind_speed"]), lam=float(base["lam"]))
    b.set_mode(_abi.OCEAN_MODE_JACOBIAN)
    for i, t in enumerate(g["times"]):
        b.compute_waves(float(t))
        d, q = b.read_maps()
        w = g[f"w{i}"]
        assert np.abs(d[0][..., 3] - w).max() <= TOL * np.abs(w).max()
        for c in range(3):
            m = max(float(np.abs(base[f"disp{i}"][..., c]).max()), 1e-30)
            assert float(np.abs(d[0][..., c] - base[f"disp{i}"][..., c]).max()) <= TOL * m
        assert max(chan_err(q[0], base[f"nrm{i}"])) <= TOL          # the seven reference fields are the FULL7 fixture's
    b.close()
