"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY (ctypes loader for the C oracle
plus an independent numpy/scipy float64 restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (watersurfacerendering_amd/) never does.

PARITY UNPINNED: the reference (kentril0/WaterSurfaceRendering) has no tests
or golden vectors for WSTessendorf and cannot be built on this image (needs
FFTW 3.3.10 + glm, neither present).  See oracle/ocean_oracle.c header.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libocean_oracle.so")

MODE_FULL7, MODE_CHOPPY5, MODE_HEIGHT1, MODE_JACOBIAN = 0, 1, 2, 3
FFT_F32, FFT_F64, FFT_F32_TEAM, FFT_FFTW, FFT_EXTERNAL = 0, 1, 2, 3, 4     # see ocean_oracle.c


def build(force: bool = False) -> str:
    """Compile the C oracle (gcc) if needed; returns the .so path."""
    src = [os.path.join(_HERE, f) for f in ("ocean_oracle.c", "fft_impl.inc", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "libocean_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
FFT_CB = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_void_p)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        P = C.c_void_p
        L.oracle_create.restype = P
        L.oracle_create.argtypes = [C.c_uint32, C.c_float]
        L.oracle_destroy.argtypes = [P]
        L.oracle_set_tile_size.argtypes = [P, C.c_uint32]
        L.oracle_set_tile_size.restype = C.c_int
        for name in ("tile_length", "wind_speed", "animation_period", "phillips_const", "lambda", "damping"):
            getattr(L, "oracle_set_" + name).argtypes = [P, C.c_float]
            getattr(L, "oracle_set_" + name).restype = None
        L.oracle_set_wind_direction.argtypes = [P, C.c_float, C.c_float]
        L.oracle_set_wind_direction.restype = None
        L.oracle_set_dispersion.argtypes = [P, C.c_int, C.c_float]
        L.oracle_set_dispersion.restype = None
        L.oracle_prepare.argtypes = [P, C.c_uint64, C.c_void_p]
        L.oracle_prepare.restype = C.c_int
        L.oracle_compute_waves.argtypes = [P, C.c_float, C.c_int, C.c_int]
        L.oracle_compute_waves.restype = C.c_float
        for name in ("min_height", "max_height", "base_freq"):
            getattr(L, "oracle_" + name).argtypes = [P]
            getattr(L, "oracle_" + name).restype = C.c_float
        for name in ("displacements", "normals", "h0", "h0_conj", "omega", "kvec", "kunit", "xi"):
            getattr(L, "oracle_" + name).argtypes = [P]
            getattr(L, "oracle_" + name).restype = C.POINTER(C.c_float)
        L.oracle_field.argtypes = [P, C.c_int]
        L.oracle_field.restype = C.POINTER(C.c_float)
        L.oracle_wind.argtypes = [P, C.POINTER(C.c_float)]
        L.oracle_gauss_pair.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.oracle_num_threads.restype = C.c_int
        L.oracle_set_num_threads.argtypes = [C.c_int]
        L.oracle_set_num_threads.restype = None
        L.oracle_fftw_available.restype = C.c_int
        L.oracle_set_external_fft.argtypes = [P, FFT_CB, C.c_void_p]
        L.oracle_set_external_fft.restype = None
        L.oracle_stage_ms.argtypes = [P]
        L.oracle_stage_ms.restype = C.POINTER(C.c_double)
        L.oracle_fft2d_f32.argtypes = [C.c_int, C.c_void_p]
        L.oracle_fft2d_f64.argtypes = [C.c_int, C.c_void_p]
        _lib = L
    return _lib


DEFAULTS = dict(length=1000.0, wind=(1.0, 1.0), wind_speed=30.0, anim_period=200.0,
                phillips_a=3e-7, damping=0.1, lam=-1.0, dispersion=(0, 0.0))
DISPERSION_DEEP, DISPERSION_FINITE_DEPTH, DISPERSION_CAPILLARY = 0, 1, 2    # WSTessendorf.h:290-293, 301-304, 312-315


class Oracle:
    """Mirror of the reference class surface (WSTessendorf.h:58-122) over the C oracle."""

    def __init__(self, n: int = 512, length: float = 1000.0, **kw):
        self._L = lib()
        self._h = self._L.oracle_create(n, length)
        if not self._h:
            raise MemoryError("oracle_create failed")
        self.n = n
        p = dict(DEFAULTS)
        p.update(kw)
        self._L.oracle_set_wind_direction(self._h, *map(float, p["wind"]))
        self._L.oracle_set_wind_speed(self._h, p["wind_speed"])
        self._L.oracle_set_animation_period(self._h, p["anim_period"])
        self._L.oracle_set_phillips_const(self._h, p["phillips_a"])
        self._L.oracle_set_damping(self._h, p["damping"])
        self._L.oracle_set_lambda(self._h, p["lam"])
        self._L.oracle_set_dispersion(self._h, int(p["dispersion"][0]), float(p["dispersion"][1]))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.oracle_destroy(self._h)
            self._h = None

    def set_lambda(self, lam: float):
        self._L.oracle_set_lambda(self._h, lam)

    def prepare(self, seed: int = 0, xi: np.ndarray | None = None):
        ptr = None
        if xi is not None:
            xi = np.ascontiguousarray(xi, dtype=np.float32).reshape(self.n, self.n, 2)
            ptr = xi.ctypes.data_as(C.c_void_p)
        rc = self._L.oracle_prepare(self._h, seed, ptr)
        if rc:
            raise RuntimeError(f"oracle_prepare rc={rc}")

    def _arr(self, name: str, comps: int) -> np.ndarray:
        p = getattr(self._L, "oracle_" + name)(self._h)
        shape = (self.n, self.n, comps) if comps > 1 else (self.n, self.n)
        return np.ctypeslib.as_array(p, shape=shape)

    h0 = property(lambda s: s._arr("h0", 2))
    h0_conj = property(lambda s: s._arr("h0_conj", 2))
    omega = property(lambda s: s._arr("omega", 1))
    kvec = property(lambda s: s._arr("kvec", 2))
    kunit = property(lambda s: s._arr("kunit", 2))
    xi = property(lambda s: s._arr("xi", 2))
    min_height = property(lambda s: s._L.oracle_min_height(s._h))
    max_height = property(lambda s: s._L.oracle_max_height(s._h))
    base_freq = property(lambda s: s._L.oracle_base_freq(s._h))

    def use_pocketfft(self, workers: int | None = None):
        """Stage D of fft=FFT_EXTERNAL frames: scipy's pocketfft (complex64, in place) on `workers` threads."""
        import scipy.fft
        workers = workers or os.cpu_count()

        def cb(ptr, nfields, n, _user):
            a = np.ctypeslib.as_array(ptr, shape=(nfields, n, n, 2)).view(np.complex64)[..., 0]
            out = scipy.fft.ifft2(a, axes=(1, 2), norm="forward", workers=workers, overwrite_x=True)
            if out.ctypes.data != a.ctypes.data:
                a[...] = out

        self._fft_cb = FFT_CB(cb)            # keep the trampoline alive
        self._L.oracle_set_external_fft(self._h, self._fft_cb, None)

    @property
    def stage_ms(self):
        """Wall time of the last frame's stages: spectra (A-C), transforms (D), pack (E-F), normalise (G)."""
        p = self._L.oracle_stage_ms(self._h)
        return dict(spectra=p[0], transforms=p[1], pack=p[2], normalise=p[3])

    @property
    def wind(self):
        w = (C.c_float * 2)()
        self._L.oracle_wind(self._h, w)
        return np.float32(w[0]), np.float32(w[1])

    def field(self, f: int) -> np.ndarray:
        p = self._L.oracle_field(self._h, f)
        a = np.ctypeslib.as_array(p, shape=(self.n, self.n, 2))
        return a[..., 0] + 1j * a[..., 1]

    def compute_waves(self, t: float, mode: int = MODE_FULL7, fft: int = FFT_F64, copy: bool = True):
        """Returns (A, disp[n,n,4], normal[n,n,4])."""
        amp = self._L.oracle_compute_waves(self._h, C.c_float(t), mode, fft)
        d, nr = self._arr("displacements", 4), self._arr("normals", 4)
        if copy:
            d, nr = d.copy(), nr.copy()
        return float(amp), d, nr


def gauss_xi(seed: int, n: int) -> np.ndarray:
    """The counter-based N(0,1) draws the C oracle (and the HIP library) use."""
    L = lib()
    out = np.empty((n * n, 2), dtype=np.float32)
    re, im = C.c_float(), C.c_float()
    for i in range(n * n):
        L.oracle_gauss_pair(seed, i, C.byref(re), C.byref(im))
        out[i, 0], out[i, 1] = re.value, im.value
    return out.reshape(n, n, 2)


def gauss_xi_numpy(seed: int, n: int) -> np.ndarray:
    """Vectorised numpy restatement of oracle_gauss_pair (splitmix64 + Box-Muller in double)."""
    idx = np.arange(n * n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u1 = ((z >> np.uint64(40)).astype(np.float64) + 1.0) / 16777216.0
    u2 = ((z >> np.uint64(8)) & np.uint64(0xFFFFFF)).astype(np.float64) / 16777216.0
    r = np.sqrt(-2.0 * np.log(u1))
    a = 2.0 * np.pi * u2
    return np.stack([(r * np.cos(a)).astype(np.float32), (r * np.sin(a)).astype(np.float32)],
                    axis=-1).reshape(n, n, 2)


# ---------------------------------------------------------------------------
# Independent numpy restatement (float32 front end, float64 pocketfft).
# Follows the same reference lines as the C oracle but shares no code with it;
# tests/test_oracle.py requires the two to agree.

def numpy_prepare(n, xi, length=1000.0, wind=(1.0, 1.0), wind_speed=30.0, anim_period=200.0,
                  phillips_a=3e-7, damping=0.1, dispersion=(0, 0.0), **_):
    f32 = np.float32
    wx, wy = f32(wind[0]), f32(wind[1])
    inv = f32(1.0) / np.sqrt(wx * wx + wy * wy, dtype=f32)
    wx, wy = f32(wx * inv), f32(wy * inv)                                  # WSTessendorf.cpp:476-479
    v = f32(max(1e-4, wind_speed))                                          # :481-484
    base = f32(np.float64(f32(2.0)) * np.pi / np.float64(f32(anim_period)))  # :486-490
    idx = np.arange(n, dtype=np.int32)
    k1 = (np.pi * (f32(2.0) * idx.astype(f32) - f32(n)).astype(np.float64)
          / np.float64(f32(length))).astype(f32)                           # :76-79
    kx = np.broadcast_to(k1[None, :], (n, n)).astype(f32)
    kz = np.broadcast_to(k1[:, None], (n, n)).astype(f32)
    d = (kx * kx + kz * kz).astype(f32)
    klen = np.sqrt(d, dtype=f32)
    ok = klen > f32(1e-5)
    with np.errstate(divide="ignore", invalid="ignore"):
        rinv = (f32(1.0) / np.sqrt(d, dtype=f32)).astype(f32)
        ux = np.where(ok, kx * rinv, f32(0)).astype(f32)                    # .h:133-136
        uz = np.where(ok, kz * rinv, f32(0)).astype(f32)
        k2 = (klen * klen).astype(f32)
        k4 = (k2 * k2).astype(f32)
        cf = (ux * wx + uz * wy).astype(f32)
        cf = (cf * cf).astype(f32)
        lw = f32(f32(v * v) / f32(9.81))
        l2 = f32(lw * lw)
        e1 = np.exp((f32(-1.0) / (k2 * l2)).astype(f32), dtype=f32)
        e2 = np.exp(((-k2) * f32(damping) * f32(damping)).astype(f32), dtype=f32)
        ph = (((f32(phillips_a) * e1).astype(f32) / k4).astype(f32) * cf).astype(f32) * e2  # .h:249-263
        ph = ph.astype(f32)
        s = f32(1.0) / np.sqrt(f32(2.0), dtype=f32)
        sp = np.sqrt(ph, dtype=f32)
        h0r = np.where(ok, ((s * xi[..., 0]).astype(f32) * sp).astype(f32), f32(0))   # .h:237-243
        h0i = np.where(ok, ((s * xi[..., 1]).astype(f32) * sp).astype(f32), f32(0))
        gk = (f32(9.81) * klen).astype(f32)
        if dispersion[0] == 1:      # .h:301-304, evaluated in double and rounded once (tanhf is libm-dependent)
            w = np.sqrt(gk.astype(np.float64) * np.tanh(klen.astype(np.float64) * np.float64(f32(dispersion[1])))).astype(f32)
        elif dispersion[0] == 2:    # .h:312-315, float, left to right
            ll = f32(dispersion[1])
            w = np.sqrt((gk * (f32(1.0) + (((klen * klen).astype(f32) * ll).astype(f32) * ll).astype(f32)).astype(f32)).astype(f32), dtype=f32)
        else:                       # .h:290-293
            w = np.sqrt(gk, dtype=f32)
        om = np.where(ok, (np.floor((w / base).astype(f32)) * base).astype(f32), f32(0))      # .h:284-287
    return dict(kx=kx, kz=kz, ux=ux, uz=uz, h0=(h0r + 1j * h0i).astype(np.complex64), omega=om.astype(f32))


def numpy_compute_waves(prep, t, lam=-1.0, jacobian=False):
    """Closed form of WSTessendorf.cpp:284-455 with float64 FFTs.  jacobian=True adds the
    reference's COMPUTE_JACOBIAN intent (.cpp:330-335, 421-428): displacement.w = (1 + lam dxDx)(1 + lam dzDz) -
    (lam dxDz)(lam dzDx)."""
    import scipy.fft as sfft
    f32 = np.float32
    n = prep["kx"].shape[0]
    wt = (prep["omega"] * f32(t)).astype(f32)                               # .h:267 single fp32 multiply
    c, s = np.cos(wt.astype(np.float64)), np.sin(wt.astype(np.float64))
    c, s = c.astype(f32), s.astype(f32)
    h0 = prep["h0"]
    hr = (f32(2.0) * ((h0.real * c).astype(f32) - (h0.imag * s).astype(f32)).astype(f32)).astype(np.float64)
    kx, kz, ux, uz = (prep[k].astype(np.float64) for k in ("kx", "kz", "ux", "uz"))
    sign = (1.0 - 2.0 * ((np.arange(n)[:, None] + np.arange(n)[None, :]) & 1)).astype(np.float64)

    def bre(x):  # Re of the unnormalised backward DFT
        return sfft.ifft2(x, norm="forward").real

    h = sign * bre(hr)
    dx = sign * bre(-1j * ux * hr)
    dz = sign * bre(-1j * uz * hr)
    sx = sign * bre(1j * kx * hr)
    sz = sign * bre(1j * kz * hr)
    dxdx = sign * bre(kx * ux * hr)
    dzdz = sign * bre(kz * uz * hr)
    hmin = min(float(h.min()), float(np.finfo(np.float32).max))
    hmax = max(float(h.max()), float(np.finfo(np.float32).tiny))           # .cpp:289 quirk
    amp = max(abs(hmin), abs(hmax))
    w = np.ones_like(h)
    if jacobian:
        dzdx = sign * bre(kz * ux * hr)          # i kz * (-i ux) = kz ux
        dxdz = sign * bre(kx * uz * hr)
        w = (1.0 + lam * dxdx) * (1.0 + lam * dzdz) - (lam * dxdz) * (lam * dzdx)
    disp = np.stack([lam * dx, h / amp, lam * dz, w], axis=-1)
    nrm = np.stack([sx, sz, dxdx, dzdz], axis=-1)
    return amp, disp, nrm, hmin, hmax
