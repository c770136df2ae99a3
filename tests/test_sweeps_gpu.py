"""Randomised GPU sweeps (fixed seeds): parameters far from the reference's defaults, every mode, every precision.

  * fp32 path against the float64-FFT oracle over random tile lengths (50 m ... 8 km), winds (1e-5 ... 80 m/s: the
    clamp of WSTessendorf.cpp:481-484 included), Phillips constants, dampings (0 included), animation periods,
    positive and negative lambda, the four modes and times up to 2000 s: 1e-5 of every channel's maximum, omega
    bit-exact.  (This sweep found the Jacobian mode's cross derivative inheriting the rounding error of a height a
    thousand times its size; it now goes through its transform amplified to the height's magnitude.)
  * half2 intermediates against the fp32 path of the same library: 1e-3, never a non-finite value.
  * every combination of mode x precision x depth x batch on one long-lived context equals a fresh context bit for bit.
"""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel_errors(got, ref):
    out = []
    for c in range(4):
        m = float(np.abs(ref[..., c]).max())
        e = float(np.abs(got[..., c].astype(np.float64) - ref[..., c]).max())
        if m == 0.0:
            assert e == 0.0, "non-zero where the oracle is exactly zero"
        else:
            out.append(e / m)
    return out


def test_fp32_parameter_sweep_against_oracle():
    import watersurfacerendering_amd as W
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    worst = 0.0
    for n in (16, 32, 64, 128, 256):
        for _ in range(10):
            seed = int(rng.integers(1, 1 << 30))
            L = float(rng.choice([50.0, 250.0, 1000.0, 8000.0]))
            wind = (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)) or 0.3)
            V = float(rng.choice([1e-5, 2.0, 30.0, 80.0])); A = float(10 ** rng.uniform(-9, -5))
            l = float(rng.choice([0.0, 0.1, 2.0])); T = float(rng.choice([20.0, 200.0, 2000.0]))
            lam = float(rng.uniform(-3, 1)); mode = int(rng.integers(0, 4))
            b = W.OceanBatch(n, 1, 0)
            b.set_params(tile_length=L, wind_dir_x=wind[0], wind_dir_y=wind[1], wind_speed=V, phillips_const=A, damping=l, anim_period=T, lambda_=lam)
            b.set_mode(mode); b.prepare(seed)
            o = O.Oracle(n, L, wind=wind, wind_speed=V, phillips_a=A, damping=l, anim_period=T, lam=lam)
            o.prepare(xi=b.read_xi(0))
            assert np.array_equal(b.read_spectrum(0)[1], o.omega), (n, seed, L, V, T)
            for t in (0.0, float(rng.uniform(-10, 2000))):
                ag = float(b.compute_waves(t)[0]); d, q = b.read_maps()
                ao, do, no = o.compute_waves(t, mode=mode, fft=O.FFT_F64)
                assert abs(ag - ao) <= 2e-6 * ao, (n, seed, mode, t, ag, ao)
                errs = _rel_errors(d[0], do) + _rel_errors(q[0], no)
                assert max(errs) <= 1e-5, (n, seed, L, wind, V, A, l, T, lam, mode, t, errs)
                worst = max(worst, max(errs))
            b.close()
    assert worst > 0.0


def test_half2_intermediates_sweep_against_the_fp32_path():
    import watersurfacerendering_amd as W
    rng = np.random.default_rng(2)
    for n in (32, 64, 128, 512):
        for _ in range(10 if n <= 128 else 3):
            seed = int(rng.integers(1, 1 << 30))
            p = dict(tile_length=float(rng.choice([100.0, 250.0, 1000.0, 4000.0])), wind_dir_x=float(rng.uniform(-1, 1)),
                     wind_dir_y=float(rng.uniform(0.05, 1)), wind_speed=float(rng.uniform(2, 60)), phillips_const=float(10 ** rng.uniform(-8, -5)),
                     damping=float(rng.uniform(0.0, 1.0)), lambda_=float(rng.uniform(-3, -0.2)))
            mode = int(rng.choice([0, 3]))
            a = W.OceanBatch(n, 1, 0); a.set_params(**p); a.set_mode(mode); a.prepare(seed)
            b = W.OceanBatch(n, 1, 0); b.set_params(**p); b.set_mode(mode); b.set_intermediate_precision(16); b.prepare(seed)
            for t in (0.0, float(rng.uniform(0, 500))):
                a.compute_waves(t); b.compute_waves(t)
                d1, q1 = a.read_maps(); d2, q2 = b.read_maps()
                assert np.all(np.isfinite(d2)) and np.all(np.isfinite(q2)), (n, seed, p, mode, t)
                errs = _rel_errors(d2[0], d1[0].astype(np.float64)) + _rel_errors(q2[0], q1[0].astype(np.float64))
                assert max(errs) <= 1e-3, (n, seed, p, mode, t, errs)
            a.close(); b.close()


def test_every_mode_precision_depth_combination_on_one_context():
    import watersurfacerendering_amd as W
    for n, tiles, period in [(64, 5, 200.0), (512, 2, 200.0), (64, 2, 4.0e5)]:     # the last: dispersion multiples beyond 16 bits (fp32 omega variants)
        b = W.OceanBatch(n, tiles, 0)
        b.set_params(anim_period=period)
        for (bits, hbits), mode, depth in itertools.product(((32, 32), (16, 32), (16, 16), (32, 16), (32, 32)), (0, 3, 1, 2, 0), (1, 3)):
            b.set_intermediate_precision(bits); b.set_spectrum_precision(hbits); b.set_mode(mode); b.set_pipeline_depth(depth)
            b.prepare(123)
            for j in range(4):
                b.compute_waves_async(0.2 * j)
            b.compute_waves_async(1.7); b.synchronize()
            d, q = b.read_maps(tiles - 1, 1)
            assert np.all(np.isfinite(d)) and np.all(np.isfinite(q)), (n, period, bits, hbits, mode, depth)
            f = W.OceanBatch(n, 1, 0); f.set_params(anim_period=period); f.set_intermediate_precision(bits); f.set_spectrum_precision(hbits)
            f.set_mode(mode); f.prepare(123 + tiles - 1)
            f.compute_waves(1.7); d2, q2 = f.read_maps(); f.close()
            assert np.array_equal(d, d2) and np.array_equal(q, q2), (n, tiles, period, bits, hbits, mode, depth)
        b.close()


def test_batch_with_different_parameters_per_tile_and_resizes():
    """A batch whose tiles differ in EVERY parameter (tile length, wind, Phillips constant, damping, lambda, time offset)
    equals single-tile contexts with the same parameters bit for bit, in the fp32, half2 and Jacobian forms; and a context
    keeps its mode / precision / depth settings across SetTileSize."""
    import watersurfacerendering_amd as W
    rng = np.random.default_rng(11)
    n, tiles, seed = 128, 4, 900
    per = [dict(tile_length=float(rng.choice([120.0, 1000.0, 3000.0])), wind_dir_x=float(rng.uniform(-1, 1)), wind_dir_y=float(rng.uniform(0.1, 1)),
                wind_speed=float(rng.uniform(3, 50)), phillips_const=float(10 ** rng.uniform(-8, -6)), damping=float(rng.uniform(0, 0.5)),
                lambda_=float(rng.uniform(-2.5, -0.3))) for _ in range(tiles)]
    offs = np.array([0.0, -1.5, 40.0, 1234.5], np.float32)
    for bits, mode in ((32, 0), (16, 0), (32, 3), (16, 3)):
        b = W.OceanBatch(n, tiles, 0)
        for i, p in enumerate(per):
            b.set_params(tile=i, **p)
        b.set_intermediate_precision(bits); b.set_mode(mode); b.set_time_offsets(offs)
        b.prepare(seed)
        amps = b.compute_waves(2.0)
        d, q = b.read_maps()
        for i, p in enumerate(per):
            s = W.OceanBatch(n, 1, 0); s.set_params(**p); s.set_intermediate_precision(bits); s.set_mode(mode); s.prepare(seed + i)
            a1 = s.compute_waves(float(np.float32(2.0) + offs[i])); d1, q1 = s.read_maps(); s.close()
            assert np.array_equal(d[i], d1[0]) and np.array_equal(q[i], q1[0]) and amps[i] == a1[0], (bits, mode, i)
        # resize: settings survive, results equal a fresh context of the new size
        b.set_tile_size(64)
        b.prepare(seed)
        b.compute_waves(2.0)
        d64, q64 = b.read_maps(tiles - 1, 1)
        f = W.OceanBatch(64, 1, 0); f.set_params(**per[-1]); f.set_intermediate_precision(bits); f.set_mode(mode); f.prepare(seed + tiles - 1)
        f.compute_waves(float(np.float32(2.0) + offs[-1])); df, qf = f.read_maps(); f.close()
        assert np.array_equal(d64, df) and np.array_equal(q64, qf), (bits, mode, "after resize")
        b.close()
