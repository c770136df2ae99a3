"""Generates tests/golden/*.npz.

The reference (kentril0/WaterSurfaceRendering) holds no golden vectors for this
path and cannot be built or imported here (C++ needing FFTW + glm), so these
fixtures are produced by the CPU ORACLE (oracle/ocean_oracle.c, float64 FFT
mode) -- they pin the oracle against regressions and give the GPU tests
fixed inputs/expected outputs; they are NOT reference outputs ("parity
unpinned").

    python tests/golden/make_golden.py

ocean_n*.npz: xi (n,n,2) f32 input draws, params, per t: amp, min, max, disp, nrm (f32).
jacobian_n*.npz: displacement.w of OCEAN_MODE_JACOBIAN for the same inputs (per t: w (n,n) f32).
sampled_n*.npz (n = 256, 512, 1024): seed, per t: amp, min, max, per-channel mean and max of
|value|, and 1024 texels sampled by a fixed LCG (index list included).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
TIMES = [0.0, 1.5, 7.25, 1000.0]
CASES = {
    "default": dict(),
    "alt": dict(length=250.0, wind=(1.0, 0.0), wind_speed=10.0, lam=-2.0),
}


def main():
    for n in (16, 32, 64):
        for name, kw in CASES.items():
            seed = 0x5EED0000 + n
            xi = O.gauss_xi_numpy(seed, n)
            kw2 = dict(kw)
            length = kw2.pop("length", 1000.0)
            o = O.Oracle(n, length, **kw2)
            o.prepare(xi=xi)
            out = {"xi": xi, "seed": np.uint64(seed), "n": np.int32(n), "times": np.array(TIMES, np.float32),
                   "length": np.float32(length),
                   "wind": np.array(kw.get("wind", (1.0, 1.0)), np.float32),
                   "wind_speed": np.float32(kw.get("wind_speed", 30.0)),
                   "lam": np.float32(kw.get("lam", -1.0)),
                   "h0": o.h0.copy(), "omega": o.omega.copy()}
            for i, t in enumerate(TIMES):
                amp, d, q = o.compute_waves(t, fft=O.FFT_F64)
                out[f"amp{i}"] = np.float32(amp)
                out[f"min{i}"] = np.float32(o.min_height)
                out[f"max{i}"] = np.float32(o.max_height)
                out[f"disp{i}"] = d.astype(np.float32)
                out[f"nrm{i}"] = q.astype(np.float32)
            np.savez_compressed(os.path.join(HERE, f"ocean_n{n}_{name}.npz"), **out)
            print("wrote", f"ocean_n{n}_{name}.npz")


def lcg_indices(n, count=1024, state=0x2545F491):
    """Texel sample list of SURVEY.md 8c: a fixed 32-bit LCG (Numerical Recipes constants), indices mod n*n."""
    idx = np.empty(count, np.int64)
    for i in range(count):
        state = (1664525 * state + 1013904223) & 0xFFFFFFFF
        idx[i] = state % (n * n)
    return idx


def sampled():
    """SURVEY.md 8c, larger sizes: A/min/max, per-channel mean |value| and 1024 sampled texels per time.
    The draws are not stored: xi = gauss_xi(seed) is part of the oracle and of the library (same generator)."""
    for n in (256, 512, 1024):
        seed = 0x5EED0000 + n
        xi = O.gauss_xi_numpy(seed, n)
        o = O.Oracle(n)
        o.prepare(xi=xi)
        idx = lcg_indices(n)
        out = {"seed": np.uint64(seed), "n": np.int32(n), "times": np.array(TIMES, np.float32), "index": idx}
        for i, t in enumerate(TIMES):
            amp, d, q = o.compute_waves(t, fft=O.FFT_F64)
            d, q = d.reshape(-1, 4), q.reshape(-1, 4)
            out[f"amp{i}"] = np.float32(amp)
            out[f"min{i}"] = np.float32(o.min_height)
            out[f"max{i}"] = np.float32(o.max_height)
            out[f"meanabs{i}"] = np.concatenate([np.abs(d).mean(0, dtype=np.float64), np.abs(q).mean(0, dtype=np.float64)])
            out[f"maxabs{i}"] = np.concatenate([np.abs(d).max(0), np.abs(q).max(0)]).astype(np.float32)
            out[f"disp{i}"] = d[idx].astype(np.float32)
            out[f"nrm{i}"] = q[idx].astype(np.float32)
        np.savez_compressed(os.path.join(HERE, f"sampled_n{n}_default.npz"), **out)
        print("wrote", f"sampled_n{n}_default.npz")


def jacobian():
    """OCEAN_MODE_JACOBIAN: displacement.w = the Jacobian of the horizontal displacement, per the
    intent of WSTessendorf.cpp:330-335,421-428, for the inputs of the ocean_n*.npz fixtures (same seeds and parameters)."""
    for n in (16, 32, 64):
        for name, kw in CASES.items():
            seed = 0x5EED0000 + n
            xi = O.gauss_xi_numpy(seed, n)
            kw2 = dict(kw)
            length = kw2.pop("length", 1000.0)
            o = O.Oracle(n, length, **kw2)
            o.prepare(xi=xi)
            out = {"seed": np.uint64(seed), "n": np.int32(n), "times": np.array(TIMES, np.float32), "inputs": f"ocean_n{n}_{name}.npz"}
            for i, t in enumerate(TIMES):
                _, d, _ = o.compute_waves(t, mode=O.MODE_JACOBIAN, fft=O.FFT_F64)
                out[f"w{i}"] = d[..., 3].astype(np.float32)
            np.savez_compressed(os.path.join(HERE, f"jacobian_n{n}_{name}.npz"), **out)
            print("wrote", f"jacobian_n{n}_{name}.npz")


if __name__ == "__main__":
    main()
    sampled()
    jacobian()
