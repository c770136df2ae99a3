"""CPU tests of the oracle (test infrastructure): pins its DFT to the
mathematical definition, its front end to an independent numpy restatement of
the same reference lines, and checks the committed golden fixtures.

PARITY UNPINNED vs the reference itself: it has no tests/vectors and cannot be
built here (FFTW 3.3.10 + glm absent); see oracle/ocean_oracle.c header."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def naive_backward_dft2(x):
    """B[X](p,q) = sum_{m,n} X(m,n) exp(+2 pi i (p m + q n)/N), float64, O(N^4)."""
    n = x.shape[0]
    w = np.exp(2j * np.pi * np.outer(np.arange(n), np.arange(n)) / n)
    return w @ x @ w.T


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_fft_matches_naive_dft(n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    ref = naive_backward_dft2(x)
    d = np.ascontiguousarray(np.stack([x.real, x.imag], -1), dtype=np.float64)
    assert O.lib().oracle_fft2d_f64(n, d.ctypes.data_as(C.c_void_p)) == 0
    got = d[..., 0] + 1j * d[..., 1]
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    f = np.ascontiguousarray(np.stack([x.real, x.imag], -1), dtype=np.float32)
    assert O.lib().oracle_fft2d_f32(n, f.ctypes.data_as(C.c_void_p)) == 0
    got32 = f[..., 0] + 1j * f[..., 1]
    assert np.abs(got32 - ref).max() <= 2e-6 * np.abs(ref).max()


@pytest.mark.parametrize("n", [64, 256, 1024])
def test_fft_matches_pocketfft(n):
    import scipy.fft as sfft
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    ref = sfft.ifft2(x, norm="forward")          # unnormalised backward transform (FFTW_BACKWARD)
    d = np.ascontiguousarray(np.stack([x.real, x.imag], -1), dtype=np.float64)
    O.lib().oracle_fft2d_f64(n, d.ctypes.data_as(C.c_void_p))
    assert np.abs(d[..., 0] + 1j * d[..., 1] - ref).max() <= 1e-12 * np.abs(ref).max()
    f = np.ascontiguousarray(np.stack([x.real, x.imag], -1), dtype=np.float32)
    O.lib().oracle_fft2d_f32(n, f.ctypes.data_as(C.c_void_p))
    assert np.abs(f[..., 0] + 1j * f[..., 1] - ref).max() <= 3e-6 * np.abs(ref).max()


def test_fft_rejects_non_pow2():
    d = np.zeros((12, 12, 2))
    assert O.lib().oracle_fft2d_f64(12, d.ctypes.data_as(C.c_void_p)) != 0


def test_gauss_rng_c_equals_numpy_and_is_standard_normal():
    a = O.gauss_xi(77, 32)
    b = O.gauss_xi_numpy(77, 32)
    assert np.array_equal(a, b)
    big = O.gauss_xi_numpy(5, 256)
    assert abs(big.mean()) < 0.01 and abs(big.std() - 1.0) < 0.01
    assert abs(np.mean(big[..., 0] * big[..., 1])) < 0.01
    assert not np.array_equal(O.gauss_xi_numpy(5, 16), O.gauss_xi_numpy(6, 16))


ALT = dict(length=250.0, wind=(1.0, 0.0), wind_speed=10.0, lam=-2.0)


@pytest.mark.parametrize("n", [16, 64, 128])
@pytest.mark.parametrize("alt", [False, True])
def test_oracle_matches_independent_numpy_restatement(n, alt):
    kw = dict(ALT) if alt else {}
    xi = O.gauss_xi_numpy(11 + n, n)
    length = kw.pop("length", 1000.0)
    o = O.Oracle(n, length, **kw)
    o.prepare(xi=xi)
    prep = O.numpy_prepare(n, xi, length=length, **{k: v for k, v in kw.items() if k != "lam"})
    assert np.array_equal(o.omega, prep["omega"])
    assert np.array_equal(o.kvec[..., 0], prep["kx"]) and np.array_equal(o.kvec[..., 1], prep["kz"])
    assert np.array_equal(o.kunit[..., 0], prep["ux"])
    h0 = o.h0[..., 0] + 1j * o.h0[..., 1]
    assert np.abs(h0 - prep["h0"]).max() <= 5e-7 * np.abs(prep["h0"]).max()   # expf: glibc vs numpy
    # conj(h0(-k)) term of the reference equals conj(h0(k)) exactly
    assert np.array_equal(o.h0_conj[..., 0], o.h0[..., 0])
    assert np.array_equal(o.h0_conj[..., 1], -o.h0[..., 1])
    lam = kw.get("lam", -1.0)
    for t in (0.0, 1.5, 1000.0):
        amp, d, q = o.compute_waves(t, fft=O.FFT_F64)
        amp_n, d_n, q_n, mn, mx = O.numpy_compute_waves(prep, t, lam=lam)
        assert abs(amp - amp_n) <= 1e-6 * amp_n
        for c in range(4):
            assert np.abs(d[..., c] - d_n[..., c]).max() <= 1e-6 * max(np.abs(d_n[..., c]).max(), 1e-30)
            assert np.abs(q[..., c] - q_n[..., c]).max() <= 1e-6 * max(np.abs(q_n[..., c]).max(), 1e-30)
        # the float-FFT "reference shape" stays within the fp32 floor of the f64 one
        amp32, d32, q32 = o.compute_waves(t, fft=O.FFT_F32)
        for c in range(3):
            assert np.abs(d32[..., c] - d[..., c]).max() <= 2e-6 * np.abs(d[..., c]).max()


def test_animated_height_spectrum_is_real_and_minmax_quirk():
    """h~ is exactly real; with a zero spectrum the reported max is
    FLT_MIN, not 0 (WSTessendorf.cpp:289-290) and A = FLT_MIN."""
    n = 16
    o = O.Oracle(n, phillips_a=0.0)
    o.prepare(seed=3)
    amp, d, q = o.compute_waves(2.0, fft=O.FFT_F32)
    tiny = float(np.finfo(np.float32).tiny)
    assert o.max_height == pytest.approx(tiny, rel=0, abs=0)
    assert amp == pytest.approx(tiny, rel=0, abs=0)
    assert np.all(d[..., 1] == 0.0) and np.all(d[..., 3] == 1.0)


def test_modes_height1_and_choppy5():
    n = 32
    xi = O.gauss_xi_numpy(2, n)
    o = O.Oracle(n)
    o.prepare(xi=xi)
    a7, d7, q7 = o.compute_waves(1.0, mode=O.MODE_FULL7, fft=O.FFT_F32)
    a5, d5, q5 = o.compute_waves(1.0, mode=O.MODE_CHOPPY5, fft=O.FFT_F32)
    a1, d1, q1 = o.compute_waves(1.0, mode=O.MODE_HEIGHT1, fft=O.FFT_F32)
    assert a7 == a5 == a1
    assert np.array_equal(d7, d5) and np.array_equal(q7[..., :2], q5[..., :2]) and np.all(q5[..., 2:] == 0)
    assert np.array_equal(d1[..., 1], d7[..., 1]) and np.all(d1[..., 0] == 0) and np.all(q1 == 0)


def test_lambda_takes_effect_without_prepare():
    n = 16
    o = O.Oracle(n)
    o.prepare(seed=1)
    _, d1, q1 = o.compute_waves(0.5, fft=O.FFT_F32)
    o.set_lambda(-2.0)
    _, d2, q2 = o.compute_waves(0.5, fft=O.FFT_F32)
    assert np.allclose(d2[..., 0], 2 * d1[..., 0], rtol=1e-6) and np.array_equal(q1, q2)
    assert np.array_equal(d1[..., 1], d2[..., 1])


def test_animation_loops_with_the_period_and_fields_are_linear_in_xi():
    """Two properties of the reference's algorithm that hold at every size (the GPU tests repeat them at 2048^2):
    (a) the dispersion is quantised to multiples of the base frequency 2 pi / T (QDispersion, .h:284-287) so that the animation LOOPS: the frame at
        t + T is the frame at t, up to the rounding of the single fp32 product omega * t (.h:267): |omega T - 2 pi q| <= ulp(omega T) / 2, i.e. a
        phase error of at most 2^-24 * omega * T ~ 4e-5 rad at the largest omega of these tiles -> 1e-4 of a channel's maximum;
    (b) h0 = ((1/sqrt 2) xi) sqrt P (.h:237-243) is linear in xi and everything after it is linear in h0: doubling xi doubles every raw field and A,
        exactly (a power of two commutes with every fp32 rounding above the subnormal range), and leaves the normalised height as it was."""
    n = 64
    rng = np.random.default_rng(7)
    xi = rng.standard_normal((n, n, 2)).astype(np.float32)
    o = O.Oracle(n); o.prepare(xi=xi)
    period = 200.0
    for t in (0.0, 3.25):
        a0, d0, q0 = o.compute_waves(t, fft=O.FFT_F64)
        a1, d1, q1 = o.compute_waves(t + period, fft=O.FFT_F64)
        assert abs(a1 - a0) <= 1e-4 * a0
        for c in range(4):
            assert np.abs(d1[..., c] - d0[..., c]).max() <= 1e-4 * max(np.abs(d0[..., c]).max(), 1e-30)
            assert np.abs(q1[..., c] - q0[..., c]).max() <= 1e-4 * np.abs(q0[..., c]).max()
    # half a period later the frame is a different one (the property above is not vacuous)
    _, dh, _ = o.compute_waves(0.5 * period, fft=O.FFT_F64)
    _, d0, _ = o.compute_waves(0.0, fft=O.FFT_F64)
    assert np.abs(dh[..., 0] - d0[..., 0]).max() > 0.1 * np.abs(d0[..., 0]).max()
    o2 = O.Oracle(n); o2.prepare(xi=2.0 * xi)
    a, d, q = o.compute_waves(1.5, fft=O.FFT_F32)
    a2, d2, q2 = o2.compute_waves(1.5, fft=O.FFT_F32)
    tiny = 1e-30            # (values that left the normal range on the way are not exact multiples)
    assert abs(a2 - 2.0 * a) <= 1e-6 * a2
    assert np.abs(q2 - 2.0 * q).max() <= 1e-6 * np.abs(q2).max() + tiny
    assert np.abs(d2[..., [0, 2]] - 2.0 * d[..., [0, 2]]).max() <= 1e-6 * np.abs(d2[..., [0, 2]]).max() + tiny
    assert np.abs(d2[..., 1] - d[..., 1]).max() <= 1e-6


def test_reference_output_point_symmetry():
    """h~(k) is real for every k (.cpp:131-135, .h:265-275) => every output field is even or odd
    under (p,q) -> (-p,-q) mod N: height and dD/dx even, displacements and slopes odd.
    The HIP pipeline computes one half and mirrors the other (ocean_kernels.h)."""
    n = 64
    o = O.Oracle(n, **ALT_SYM)
    o.prepare(seed=5)
    for fft in (O.FFT_F64, O.FFT_F32):
        _, d, q = o.compute_waves(3.3, fft=fft)

        def mir(a):
            return np.roll(a[::-1, ::-1], (1, 1), axis=(0, 1))
        tol = 0.0 if fft == O.FFT_F64 else 2e-6
        for a, eps in ((d[..., 0], -1), (d[..., 1], 1), (d[..., 2], -1), (q[..., 0], -1), (q[..., 1], -1),
                       (q[..., 2], 1), (q[..., 3], 1)):
            assert np.abs(a - eps * mir(a)).max() <= tol * np.abs(a).max() + 1e-7 * np.abs(a).max()


ALT_SYM = dict(wind=(0.3, -1.0), wind_speed=17.0)


def test_spatial_mean_is_zero():
    """DC amplitude is zero (k = 0 branch, .cpp:138-143) => every field sums to ~0 over the tile."""
    n = 64
    o = O.Oracle(n)
    o.prepare(seed=9)
    _, d, q = o.compute_waves(3.0, fft=O.FFT_F64)
    for a in (d[..., 0], d[..., 1], d[..., 2], q[..., 0], q[..., 1]):
        assert abs(a.mean()) <= 1e-6 * np.abs(a).max()


@pytest.mark.parametrize("dispersion", [(1, 2.0), (2, 5.0)])
def test_other_dispersion_relations_c_vs_numpy(dispersion):
    """WSTessendorf.h:301-315 (defined, never called by the reference): C oracle against the numpy restatement."""
    from oracle import oracle as O
    n = 64
    xi = O.gauss_xi_numpy(11, n)
    o = O.Oracle(n, dispersion=dispersion)
    o.prepare(xi=xi)
    p = O.numpy_prepare(n, xi.reshape(n, n, 2), dispersion=dispersion)
    assert np.array_equal(o.omega.reshape(n, n), p["omega"])
    d = O.Oracle(n)
    d.prepare(xi=xi)
    assert not np.array_equal(d.omega, o.omega)
    assert np.array_equal(d.h0, o.h0)                       # only the dispersion changes


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "ocean_n*.npz"))))
def test_golden_fixtures(path):
    g = np.load(path)
    n = int(g["n"])
    o = O.Oracle(n, float(g["length"]), wind=tuple(g["wind"]), wind_speed=float(g["wind_speed"]),
                 lam=float(g["lam"]))
    o.prepare(xi=g["xi"])
    assert np.array_equal(g["xi"], O.gauss_xi_numpy(int(g["seed"]), n))
    assert np.array_equal(o.omega, g["omega"])
    assert np.abs(o.h0 - g["h0"]).max() <= 1e-6 * np.abs(g["h0"]).max()
    for i, t in enumerate(g["times"]):
        amp, d, q = o.compute_waves(float(t), fft=O.FFT_F64)
        assert abs(amp - float(g[f"amp{i}"])) <= 1e-6 * amp
        assert abs(o.min_height - float(g[f"min{i}"])) <= 1e-6 * amp
        for c in range(4):
            assert np.abs(d[..., c] - g[f"disp{i}"][..., c]).max() <= 2e-6 * max(np.abs(d[..., c]).max(), 1e-30)
            assert np.abs(q[..., c] - g[f"nrm{i}"][..., c]).max() <= 2e-6 * max(np.abs(q[..., c]).max(), 1e-30)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "sampled_n256*.npz")) + glob.glob(os.path.join(GOLDEN, "sampled_n512*.npz"))))
def test_sampled_golden_fixtures(path):
    """SURVEY.md 8c, larger sizes: amplitude, per-channel statistics and 1024 LCG-sampled texels (1024^2 is
    checked on the GPU side only, to keep the CPU suite short)."""
    g = np.load(path)
    n = int(g["n"])
    o = O.Oracle(n)
    o.prepare(xi=O.gauss_xi_numpy(int(g["seed"]), n))
    idx = g["index"]
    for i, t in enumerate(g["times"]):
        amp, d, q = o.compute_waves(float(t), fft=O.FFT_F64)
        d, q = d.reshape(-1, 4), q.reshape(-1, 4)
        assert abs(amp - float(g[f"amp{i}"])) <= 1e-6 * amp
        assert abs(o.min_height - float(g[f"min{i}"])) <= 1e-6 * amp and abs(o.max_height - float(g[f"max{i}"])) <= 1e-6 * amp
        scale = np.maximum(g[f"maxabs{i}"], 1e-30)
        mean = np.concatenate([np.abs(d).mean(0, dtype=np.float64), np.abs(q).mean(0, dtype=np.float64)])
        assert np.all(np.abs(mean - g[f"meanabs{i}"]) <= 1e-6 * scale)
        assert np.all(np.abs(d[idx] - g[f"disp{i}"]) <= 2e-6 * scale[:4])
        assert np.all(np.abs(q[idx] - g[f"nrm{i}"]) <= 2e-6 * scale[4:])


# ---------------------------------------------------------------------------
# OCEAN_MODE_JACOBIAN: the reference's COMPUTE_JACOBIAN intent
def test_jacobian_mode_restates_the_reference_intent():
    """WSTessendorf.cpp:330-335 builds dz(Dx) = i kz Dx and dx(Dz) = i kx Dz; :421-428 combines them into
    (1 + l dxDx)(1 + l dzDz) - (l dxDz)(l dzDx).  The C oracle (two separate transforms, float order of the
    reference) against the numpy closed form; the seven reference fields are untouched by the mode; the two cross
    fields coincide (kz ux = kx uz); the Jacobian is even under index inversion and tends to 1 as lambda -> 0."""
    from oracle import oracle as O
    n, t = 64, 2.5
    xi = O.gauss_xi_numpy(5, n)
    for lam in (-1.0, -1.7):
        o = O.Oracle(n, lam=lam)
        o.prepare(xi=xi)
        a, d, q = o.compute_waves(t, mode=O.MODE_JACOBIAN, fft=O.FFT_F64)
        dzdx, dxdz = o.field(7).real.copy(), o.field(8).real.copy()
        a7, d7, q7 = o.compute_waves(t, mode=O.MODE_FULL7, fft=O.FFT_F64)
        assert a == a7 and np.array_equal(d[..., :3], d7[..., :3]) and np.array_equal(q, q7)
        assert np.all(d7[..., 3] == 1.0)
        assert np.abs(dzdx - dxdz).max() <= 1e-5 * np.abs(dxdz).max()
        prep = O.numpy_prepare(n, xi)
        _, dn, _, _, _ = O.numpy_compute_waves(prep, t, lam=lam, jacobian=True)
        assert np.abs(d[..., 3] - dn[..., 3]).max() <= 2e-6 * np.abs(dn[..., 3]).max()
        w = d[..., 3]
        wm = np.roll(w[::-1, ::-1], (1, 1), axis=(0, 1))              # w at ((N-m)%N, (N-n)%N)
        assert np.abs(w - wm).max() <= 1e-5
        assert w.min() < 1.0 < w.max()
    o = O.Oracle(n, lam=-1e-4)
    o.prepare(xi=xi)
    _, d, _ = o.compute_waves(t, mode=O.MODE_JACOBIAN, fft=O.FFT_F64)
    assert np.abs(d[..., 3] - 1.0).max() < 1e-3


def test_foam_mask_follows_the_fragment_stage():
    """WaterSurfaceMesh.frag:210-212 paints white where the interpolated w is negative; with stronger choppiness more
    of the surface folds."""
    from oracle import oracle as O
    from oracle import consumer as C
    n = 64
    xi = O.gauss_xi_numpy(5, n)
    frac = []
    for lam in (-0.5, -1.5, -3.0):
        o = O.Oracle(n, lam=lam)
        o.prepare(xi=xi)
        a, d, q = o.compute_waves(1.0, mode=O.MODE_JACOBIAN, fft=O.FFT_F64)
        pos, _ = C.displace_grid(d, q, a, n, 1000.0 / 512.0, 1.0, lam)
        frac.append(float(C.foam_mask(pos).mean()))
    assert frac[0] <= frac[1] <= frac[2] and frac[2] > 0.0 and frac[0] < 0.01


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "jacobian_n*.npz"))))
def test_jacobian_golden_fixtures(path):
    """Committed displacement.w of OCEAN_MODE_JACOBIAN for the inputs of the ocean_n*.npz fixtures: the C oracle reproduces
    it, and so does the independent numpy restatement."""
    g = np.load(path)
    base = np.load(os.path.join(GOLDEN, str(g["inputs"])))
    n = int(g["n"])
    kw = dict(wind=tuple(base["wind"]), wind_speed=float(base["wind_speed"]), lam=float(base["lam"]))
    o = O.Oracle(n, float(base["length"]), **kw)
    o.prepare(xi=base["xi"])
    prep = O.numpy_prepare(n, base["xi"], length=float(base["length"]), wind=kw["wind"], wind_speed=kw["wind_speed"])
    for i, t in enumerate(g["times"]):
        _, d, _ = o.compute_waves(float(t), mode=O.MODE_JACOBIAN, fft=O.FFT_F64)
        w = g[f"w{i}"]
        assert np.abs(d[..., 3] - w).max() <= 2e-6 * np.abs(w).max()
        _, dn, _, _, _ = O.numpy_compute_waves(prep, float(t), lam=kw["lam"], jacobian=True)
        assert np.abs(dn[..., 3] - w).max() <= 1e-5 * np.abs(w).max()


def test_dispersion_is_point_symmetric_bit_for_bit():
    """omega(m, n) == omega((N-m)%N, (N-n)%N) exactly: k(N-i) = -k(i) in the reference's double-then-float evaluation
    (WSTessendorf.cpp:76-79), so |k| and the quantised dispersion (.h:284-293) agree.  The HIP z pass relies on it: one
    sincos animates a spectrum element and its point mirror."""
    from oracle import oracle as O
    for n, length, kw in [(16, 1000.0, {}), (64, 370.0, dict(wind_speed=12.0)), (256, 1000.0, {})]:
        o = O.Oracle(n, length, **kw)
        o.prepare(seed=3)
        om = np.array(o.omega)
        idx = (n - np.arange(n)) % n
        assert np.array_equal(om, om[np.ix_(idx, idx)])
    for disp, param in ((1, 40.0), (2, 1000.0)):          # the two relations the reference defines but never calls
        o = O.Oracle(64, dispersion=(disp, param))
        o.prepare(seed=3)
        om = np.array(o.omega)
        idx = (64 - np.arange(64)) % 64
        assert np.array_equal(om, om[np.ix_(idx, idx)])
