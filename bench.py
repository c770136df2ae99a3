#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X ocean synthesiser.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Launched bare with --gpus N > 1 (WORLD_SIZE unset) it starts the N rank processes itself
(`python -m torch.distributed.run`, as a child, before this process touches a GPU) and exits with
their status; it never reports an `n_gpus` different from the one asked for.

One *step* = one ComputeWaves(t) of the hot path (reference WSTessendorf.cpp:284-455) on synthetic
input: a 2048 x 2048 tile, all seven output fields, reference default parameters, xi from the
counter-based RNG with seed 0x5EED0000 + tile_index, t_j = 0.05*j.  Inputs
(h0, omega) are resident in HBM before the timed region; outputs are the two finished RGBA32F maps
in HBM (D2H read-back is not part of the metric).

The K timed steps are K asynchronous ocean_compute_waves_async calls followed by one synchronise.
With --depth 3 (default) consecutive frames rotate over three independent chains (own stream, own
intermediates, own map set); every frame is still computed in full.  --depth 1 = strictly serial
frames (also reported under extra).

roofline: per-launch durations are the kernels' own execution times (events attached to the dispatches with hipExtLaunchKernelGGL on the
launch stream, serial frames in a context of their own so that every kernel has the GPU to itself; three such passes spread over the run --
before the timed regions, right behind them, at the end -- and the one with the median frame time is reported, all three in the sidecar);
bytes are what THIS pipeline has to move (73 B/texel: 23 / 28 / 22 per launch), with the PMC-measured traffic (profiles/traffic.json) and the
rocprofv3 duration of the same kernel (profiles/kernel_stats.json) beside them while the kernel sources hash to what those were measured with.

Output: rank 0 prints ONE compact JSON line on stdout (the contract's keys + roofline / cpu_baseline / gather / timing objects, numbers and
short labels only, < 8 KB: tests/test_bench_contract.py asserts it) and writes everything else it measured -- the secondary configurations,
per-kernel tables, the CPU baseline's stage split, every explanatory note -- to bench_extra.json beside this script and to stderr.

Multi-GPU: tiles are independent, so every rank synthesises its own tile(s) with no data-path
collective ("weak" scaling, value = frames of all ranks per second).  The north-star's single RCCL
gather of the packed maps (ocean_gather_maps: ncclGather from the library's own map buffers) is
measured after the timed region on BASELINE config 5's share (8 tiles of 1024^2 per rank) and
reported under "gather": compute only, compute + gather serial, gather overlapped.

BASELINE config 5 (64 independent 1024 x 1024 tiles, 8 per GPU on 8 GPUs) as the timed workload itself:

    python bench.py --gpus 8 --size 1024 --tiles 8 --depth 2

(`config.workload` then reads "64 x 1024x1024 tiles, 8 per GPU"); the default invocation keeps BASELINE's single-GPU
headline tile (2048 x 2048) per rank and measures config 5's share in the `gather` object.

Exit status: 0 only when the line carries every measurement that was asked for; a gather
that hangs or fails at N > 1 still gets its line out (the timed region is complete by then) but the process exits 3.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x5EED0000
DT = 0.05
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (about 6.3 TB/s achievable)
# Bytes per texel each launch of THIS pipeline has to move: half-size
# intermediates, 16-bit dispersion.  roofline.achieved / frac use the library's own accounting of the measured context
# (ocean_algorithmic_bytes_per_launch); these are its values for the fp32 seven-field frame, kept for the CPU-side checks.
KERNEL_BYTES_ACTUAL = {"k_zpass": 23, "k_xpass_b": 28, "k_xpass_disp": 22}      # ("k_zpass" = the frame's first launch, whichever kernel form: k_zpass / k_zpass_c1)
FRAME_BYTES_ACTUAL = 73.0
# SURVEY.md 8d's MODEL of a plain two-pass scheme with 3.5 full-size complex intermediates (no point
# symmetry): 108 B/texel per frame.  Not the traffic of this pipeline; reported as `survey_model_*` only.
KERNEL_BYTES_SURVEY = {"k_zpass": 40, "k_xpass_b": 40, "k_xpass_disp": 28}
FRAME_BYTES_SURVEY = 108.0


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the device needs a few hundred frames (tens of ms) to reach its steady state -- 200 frames after 10
    # warm-up frames measure 60 us/frame, every later batch of 200 measures 53-54 -- and 3000 frames are 0.2 s
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--regions", type=int, default=9, help="timed regions of --steps frames each (>= 7): ms_per_step is their median, p10 / p90 beside it")
    ap.add_argument("--prewarm", type=int, default=2000, help="untimed frames before the W warm-up steps (brings the device to its steady state even when W is small: 0.1 s)")
    ap.add_argument("--size", type=int, default=2048, help="tile size N (default: the roofline config, 2048)")
    ap.add_argument("--tiles", type=int, default=1, help="independent tiles per rank per step")
    ap.add_argument("--depth", type=int, default=3, help="frame pipeline depth of the asynchronous API (1 = strictly serial frames)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="budget for EACH of the two CPU baseline samples")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL gather measurement")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary 512^2 / batched measurements")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)   # internal: the CPU baseline leg as its own process
    ap.add_argument("--check-file", default=None, help=argparse.SUPPRESS)                  # internal: GPU frames the child checks against the float64 oracle
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` -> N ranks, one per GPU
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_command(args, argv) -> list:
    """The child command that runs this script as args.gpus ranks on this node."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv) -> int:
    """Runs in a process that has NOT initialised a GPU: counts devices (no HIP context), then starts the
    ranks as a child process and relays their output and status.  Never exec()s."""
    import torch
    have = torch.cuda.device_count()        # does not create a HIP context on this image
    if have < args.gpus and os.environ.get("OCEAN_BENCH_BACKEND", "nccl") == "nccl":
        print(f"bench.py: --gpus {args.gpus} requested but this node has {have} GPU(s); refusing to report "
              f"a run on fewer GPUs than asked for", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    r = subprocess.run(launch_command(args, argv), env=env)
    return r.returncode


# ---------------------------------------------------------------------------------------------
def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


def _time_oracle(o, O, fft, budget_s, max_frames=200):
    for j in range(2):
        o.compute_waves(DT * j, fft=fft, copy=False)
    times, stages = [], []
    t_start = time.perf_counter()
    j = 0
    while True:
        t0 = time.perf_counter()
        o.compute_waves(DT * (2 + j), fft=fft, copy=False)
        times.append(time.perf_counter() - t0)
        stages.append(o.stage_ms)
        j += 1
        if (time.perf_counter() - t_start >= budget_s and j >= 5) or j >= max_frames:
            break
    order = sorted(range(len(times)), key=lambda i: times[i])
    mid = order[len(order) // 2]
    return times[mid], len(times), {k: round(v, 3) for k, v in stages[mid].items()}


def usable_cpus():
    """(CPUs this process may actually run on at once, cgroup quota or None, affinity count, nproc): the smaller of the affinity mask and
    the cgroup CPU quota -- a container may show 256 logical CPUs and be allowed 16."""
    import math
    nproc = os.cpu_count() or 1
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else nproc
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    usable = aff if quota is None else max(1, min(aff, int(math.ceil(quota))))
    return usable, quota, aff, nproc


def check_frames_against_oracle(path):
    """CPU-baseline child only (the oracle as the checker): the frames the GPU side saved -- one per 4096^2 variant of BASELINE config 4, same
    injected draws, same t -- against the float64-FFT oracle: per variant max|err| / max|channel| over the saved rows, and A."""
    import numpy as np
    from oracle import oracle as O
    z = np.load(path)
    n, t = int(z["n"]), float(z["t"])
    rows = z["rows"]
    o = O.Oracle(n)
    o.prepare(xi=z["xi"])
    amp, d, q = o.compute_waves(t, fft=O.FFT_F64)
    ref = np.concatenate([d[rows], q[rows]], axis=-1).astype(np.float64)          # [rows][n][8]
    scale = np.maximum(np.abs(np.concatenate([d, q], axis=-1)).reshape(-1, 8).max(axis=0).astype(np.float64), 1e-30)
    out = {}
    for name in [str(x) for x in z["names"]]:
        got = z["maps_" + name].astype(np.float64)
        err = (np.abs(got - ref).reshape(-1, 8).max(axis=0) / scale)
        out[name] = {"max_err_over_max_channel": float(err.max()), "per_channel": [float(e) for e in err],
                     "amplitude_rel_err": float(abs(float(z["amp_" + name]) - amp) / abs(amp))}
    return {"tile_size": n, "t": t, "rows_compared": int(len(rows)), "reference": "oracle/ocean_oracle.c with its float64 FFT (the parity target of tests/)",
            "variants": out}


def cpu_baseline(n: int, budget_s: float):
    """Two CPU figures on the host cores, same workload, same inputs (each a bounded sample of `budget_s` seconds):
      cpu_baseline         the oracle in the REFERENCE'S SHAPE (WSTessendorf.cpp:292-455): OpenMP loops around
                           seven single-threaded 2-D FFTs in parallel -- with FFTW (the reference's plans) when the
                           host has libfftw3f, else the oracle's own float FFT;
      cpu_baseline_strong  the best full-frame CPU rate measured here, NOT in the reference's shape: the same
                           element-wise stages with stage D done by every core -- scipy's pocketfft (a tuned library
                           FFT) or the oracle's own FFT work-shared by the team, whichever is faster -- and the
                           normalisation loop shared out too.  Speed-ups are quoted against this one only.
    """
    from oracle import oracle as O
    o = O.Oracle(n)
    o.prepare(seed=SEED)
    usable, quota, aff, nproc = usable_cpus()
    omp_max = int(O.lib().oracle_num_threads())
    threads = max(1, min(omp_max, usable))          # the team the reference shape runs with: the CPUs the host lets this process use at once
    O.lib().oracle_set_num_threads(threads)
    have_fftw = bool(O.lib().oracle_fftw_available())
    med, cnt, split = _time_oracle(o, O, O.FFT_FFTW if have_fftw else O.FFT_F32, budget_s)
    # strong baseline: every stage on many cores; a container may expose more logical CPUs than it may run at once
    # (cgroup quota), so the team size is searched, not assumed
    cands = {}
    sizes = sorted({t for t in (8, 16, 32, 64, 128, threads, omp_max) if t <= omp_max})
    try:
        o.use_pocketfft(os.cpu_count())
    except Exception:
        pass
    for tsz in sizes:
        O.lib().oracle_set_num_threads(tsz)
        cands[f"own FFT work-shared by a team of {tsz} OpenMP threads"] = _time_oracle(o, O, O.FFT_F32_TEAM, budget_s / (2 * len(sizes)), max_frames=40) + (tsz,)
        if getattr(o, "_fft_cb", None) is not None:
            cands[f"scipy pocketfft (complex64, in place, workers={os.cpu_count()}) + {tsz} OpenMP threads for the other stages"] = \
                _time_oracle(o, O, O.FFT_EXTERNAL, budget_s / (2 * len(sizes)), max_frames=40) + (tsz,)
    O.lib().oracle_set_num_threads(threads)
    best = min(cands, key=lambda k: cands[k][0])
    med_s, cnt_s, split_s, best_threads = cands[best]
    # BASELINE config 1: 256 x 256, height only (1 iFFT), CPU path only (plumbing)
    o1 = O.Oracle(256)
    o1.prepare(seed=SEED)
    o1.compute_waves(0.0, mode=O.MODE_HEIGHT1, fft=O.FFT_F32, copy=False)
    t1 = []
    for j in range(20):
        t0 = time.perf_counter()
        o1.compute_waves(DT * j, mode=O.MODE_HEIGHT1, fft=O.FFT_F32, copy=False)
        t1.append(time.perf_counter() - t0)
    t1.sort()
    host = {"cpu_model": _cpu_model(), "nproc": nproc, "omp_max_threads": omp_max, "cgroup_cpu_quota": quota,
            "affinity_cpus": aff, "usable_cpus": usable,
            "note": "stage D of the reference shape (7 single-threaded 2-D FFTs in omp sections) cannot use more than 7 threads"}
    fft_note = ("FFTW found on this host (libfftw3f, plans as WSTessendorf.cpp:191-232)" if have_fftw else
                "FFTW not available on this host: baseline is the oracle's own float Stockham FFT")
    ref_shape = {
        "value": 1.0 / med, "unit": "frames/s", "cores": threads, "nproc": nproc, "kind": "port",
        "cores_note": "cores = the OpenMP team this sample ran with = the CPUs usable at once (min of the affinity mask and the cgroup CPU quota); "
                      "nproc = logical CPUs the host shows",
        "fft": "fftw3f" if have_fftw else "own",
        "sample": f"{cnt} frames of the same {n}x{n} 7-field workload after 2 warm-up frames, median "
                  f"({med * 1e3:.1f} ms/frame); {fft_note}, in the reference's OpenMP shape "
                  f"(7 single-threaded 2-D FFTs in parallel; {threads} threads for the element-wise loops)",
        "stage_ms_of_the_median_frame": split,
        "gtexels_per_s": n * n / med * 1e-9,
        "host": host,
        "config1_256x256_height_only_cpu_ms": t1[len(t1) // 2] * 1e3,
    }
    strong = {
        "value": 1.0 / med_s, "unit": "frames/s", "cores": best_threads, "kind": "port",
        "sample": f"{cnt_s} full frames of the same workload, median ({med_s * 1e3:.1f} ms/frame): the oracle's pipeline with "
                  f"stage D shared out ({best}) and the normalisation loop shared out too -- not the reference's shape; the "
                  "strongest full-frame CPU figure among the candidates measured in this run",
        "stage_ms_of_the_median_frame": split_s,
        "candidates_ms_per_frame": {k: v[0] * 1e3 for k, v in cands.items()},
        "gtexels_per_s": n * n / med_s * 1e-9,
    }
    return ref_shape, strong


def cpu_child_env(environ):
    """Environment of the CPU-baseline child.  torch.distributed.run exports OMP_NUM_THREADS=1 to its ranks whenever it starts more than
    one of them (and the driver starts bench.py that way at N > 1): inherited, it would time the reference's OpenMP path on ONE thread
    and the N > 1 lines would carry a CPU figure 7-16x below the N = 1 line's.  Under the launcher the variable is dropped, so the child
    sizes its team from the usable CPUs as it does at N = 1 (`cores` in the line is what it really used)."""
    env = dict(environ, OMP_WAIT_POLICY="PASSIVE")
    if "TORCHELASTIC_RUN_ID" in env or "LOCAL_WORLD_SIZE" in env:
        env.pop("OMP_NUM_THREADS", None)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)                        # the child is no rank of anything
    return env


def cpu_baseline_isolated(n: int, budget_s: float, check_file=None):
    """The CPU baseline leg in a child process with a hard time limit: the oracle is test infrastructure running on a
    host this script knows nothing about, and nothing it does may hang or take down the GPU measurement.  With check_file the
    child also compares the GPU frames saved there with the float64 oracle (third return value)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--size", str(n), "--cpu-seconds", str(budget_s)]
    limit = 60 + 8 * budget_s
    if check_file:
        cmd += ["--check-file", check_file]
        limit += 240
    env = cpu_child_env(os.environ)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=limit, env=env)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and lines:
            d = json.loads(lines[-1])
            return d["cpu_baseline"], d["cpu_baseline_strong"], d.get("check")
        err = f"CPU baseline child exited with {r.returncode}: {r.stderr[-300:]}"
    except subprocess.TimeoutExpired:
        err = f"CPU baseline child exceeded its time limit ({limit:.0f} s)"
    return {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": err}, None, None


def save_config4_frames(W, device, path, n=4096, t=1.0, nrows=48):
    """One frame of each 4096^2 variant of BASELINE config 4 ("fp32 vs fp16 spectrum", plus the half2 intermediates) from the SAME injected
    draws, saved (a spread of whole map rows + A) for the CPU-baseline child to compare with the float64 oracle.  The draws are the
    device generator's own (read back from the fp32 context), so GPU and oracle start from identical inputs."""
    import numpy as np
    rows = np.unique(np.concatenate([np.arange(0, n, n // nrows), [n // 2 - 1, n // 2, n // 2 + 1, n - 1]])).astype(np.int64)
    variants = {"fp32": {}, "fp16_spectrum": {"h0_bits": 16}, "fp16_intermediates": {"inter_bits": 16}}
    data = {"n": n, "t": t, "rows": rows, "names": np.array(list(variants))}
    xi = None
    for name, kw in variants.items():
        b = W.OceanBatch(n, 1, device)
        if kw.get("h0_bits"):
            b.set_spectrum_precision(16)
        if kw.get("inter_bits"):
            b.set_intermediate_precision(16)
        b.prepare(SEED, xi=xi)
        if xi is None:
            xi = b.read_xi(0)[None].copy()
            data["xi"] = xi[0]
        amp = b.compute_waves(t)
        d, q = b.read_maps()
        data["maps_" + name] = np.concatenate([d[0][rows], q[0][rows]], axis=-1)
        data["amp_" + name] = float(amp[0])
        b.close()
    np.savez(path, **data)
    return path


# ---------------------------------------------------------------------------------------------
def measure_config(W, n, tiles, device, steps, warmup, h0_bits=32, depth=1, mode=0, inter_bits=32):
    b = W.OceanBatch(n, tiles, device)
    if h0_bits != 32:
        b.set_spectrum_precision(h0_bits)
    if inter_bits != 32:
        b.set_intermediate_precision(inter_bits)
    if mode:
        b.set_mode(mode)
    b.set_pipeline_depth(depth)
    b.prepare(SEED)
    ms, kern = b.time_frames(0.0, DT, warmup, steps, per_kernel=True)
    per = ms / steps * 1e-3
    names = b.kernel_names()
    own = float(b.algorithmic_bytes_per_texel) if mode in (0, 3) else None
    b.close()
    out = {"size": n, "tiles_per_step": tiles, "pipeline_depth": depth, "mode": ["FULL7", "CHOPPY5", "HEIGHT1", "JACOBIAN"][mode],
           "frames_per_s": tiles / per, "us_per_step": per * 1e6, "gtexels_per_s": n * n * tiles / per * 1e-9,
           "kernel_us": {k: v * 1e3 for k, v in zip(names, kern)}}
    if own is not None:
        out["own_bytes_per_texel"] = own
        out["own_bytes_GBps"] = own * n * n * tiles / per * 1e-9
        out["frac_of_hbm_peak"] = out["own_bytes_GBps"] / HBM_PEAK_GBPS
    return out


def measure_dropin_call(W):
    """The real drop-in call: the reference's `WSTessendorf::ComputeWaves` through the C++ adaptor of
    include/WSTessendorf.hpp -- synthesis + BOTH maps in the caller's host vectors, blocking (WaterSurfaceMesh.cpp:145-154 + 701-755) -- at
    512^2 (the reference's default size), 1024^2 (its GUI's largest) and 2048^2, with the PCIe rate the call achieves over the maps' bytes.
    tests/cpp/adaptor_demo.cpp is compiled here (g++, a few seconds) and run as a child process; sidecar only, never the headline."""
    import re
    import shutil
    import tempfile
    lib_dir = os.path.dirname(W._abi.LIB_PATH)
    tmp = tempfile.mkdtemp(prefix="ocean_dropin_")
    try:
        exe = os.path.join(tmp, "adaptor_demo")
        cc = subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "adaptor_demo.cpp"),
                             "-o", exe, "-L", lib_dir, "-locean_hip", "-Wl,-rpath," + lib_dir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"],
                            capture_output=True, text=True, timeout=300)
        if cc.returncode != 0:
            return {"error": "g++: " + cc.stderr[-300:]}
        out = {"what": "WSTessendorf::ComputeWaves(t) of include/WSTessendorf.hpp from a C++ host = ocean_compute_waves_read: the frame + both maps "
                       "into page-locked host vectors, blocking; us per call (mean of `frames` back-to-back calls), GB/s = 32 B/texel over that time"}
        for n, frames in ((512, 400), (1024, 200), (2048, 100)):
            r = subprocess.run([exe, str(n), "1.5", str(frames)], capture_output=True, text=True, timeout=300)
            m = re.search(r"read-out of both maps: ([0-9.]+) us/frame = ([0-9.]+) GB/s", r.stderr)
            out[f"{n}x{n}"] = {"us_per_call": float(m.group(1)), "pcie_GBps": float(m.group(2)), "frames": frames} if m else {"error": (r.stderr or r.stdout)[-200:]}
        return out
    except Exception as exc:  # noqa: BLE001  (an extra: its failure is reported, nothing else depends on it)
        return {"error": f"{type(exc).__name__}: {exc}"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_sync_calls(W, n, device, calls=2000):
    """Latency distribution of the synchronous ComputeWaves (the reference's call shape, WaterSurfaceMesh.cpp:145-154: one blocking call per
    frame, returns the amplitude), twice: the calling thread as the scheduler places it, and pinned to one CPU with the collector off.  The
    slow calls of the first pass sit in the ENQUEUE -- the HIP launch path on a host thread that is preempted or migrated -- not in the wait
    (tools/ubench/sync_tail.cpp from a C++ host: p95 / p50 = 1.02-1.08 unpinned, 1.01-1.02 pinned; profiles/r05_sync_tail.txt)."""
    import gc
    import numpy as np
    b = W.OceanBatch(n, 1, device)
    b.prepare(SEED)

    def one_pass():
        for j in range(100):
            b.compute_waves(DT * j)
        ts = np.empty(calls)
        for j in range(calls):
            t0 = time.perf_counter()
            b.compute_waves(DT * j)
            ts[j] = time.perf_counter() - t0
        med, p95 = float(np.median(ts) * 1e6), float(np.percentile(ts, 95) * 1e6)
        return {"median_us_per_call": med, "p95_us_per_call": p95, "p99_us_per_call": float(np.percentile(ts, 99) * 1e6), "p95_over_median": p95 / med}
    free = one_pass()
    pinned = None
    if hasattr(os, "sched_setaffinity"):
        mask = os.sched_getaffinity(0)
        was_enabled = gc.isenabled()
        try:
            os.sched_setaffinity(0, {sorted(mask)[len(mask) // 2]})
            gc.disable()
            pinned = one_pass()
        except OSError:
            pinned = None
        finally:
            os.sched_setaffinity(0, mask)
            if was_enabled:
                gc.enable()
    b.close()
    return {"size": n, "calls": calls, **free, "pinned_thread_gc_off": pinned,
            "what": "host-side wall time of ocean_compute_waves called from Python: enqueue + the frame + a poll of the frame's completion records in "
                    "host-coherent memory (no stream synchronisation; it falls back to one after 2 ms, or at once when the maps are visible "
                    "outside the context: exported, bound, or handed out)"}


def measure_consumer(W, n, device, calls=200):
    """Vertex-stage consumer: (n+1)^2 displaced vertices + normals from the maps of one frame."""
    b = W.OceanBatch(n, 1, device)
    b.prepare(SEED)
    b.compute_waves(1.0)
    L, h = W._abi.lib(), b._h
    for _ in range(10):
        L.ocean_displace_grid(h, 0, n, 1000.0 / 512.0, 1.0, -1.0)
    b.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        L.ocean_displace_grid(h, 0, n, 1000.0 / 512.0, 1.0, -1.0)
    b.synchronize()
    per = (time.perf_counter() - t0) / calls
    b.close()
    verts = (n + 1) * (n + 1)
    return {"size": n, "vertices": verts, "us_per_call": per * 1e6, "gvertices_per_s": verts / per * 1e-9,
            "what": "ocean_displace_grid back to back on one stream: bilinear REPEAT sampling of both maps, positions + normals out"}


def kernel_source_sha16() -> str:
    """Hash of the kernel sources: the committed profile summaries carry the hash they were measured with, and their
    figures are quoted in the line only while it matches (no silently stale numbers beside live ones)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "watersurfacerendering_amd", "csrc")
    for f in sorted(os.listdir(csrc)):      # the device code and the frame launcher (not the host side of the ABI, ocean_api.hip)
        if f in ("ocean_kernels.h", "fft_engine.h", "ocean_launch.h") or (f.startswith("frames_") and f.endswith(".hip")):
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def load_profiles_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return {}


def roofline_object(n, tiles, names, kern_ms, kern_ms_pipe, depth, ms_per_step, serial_us_per_step, own_bytes_per_texel,
                    kernel_bytes=None):
    """Roofline of the dominant (longest) kernel on this pipeline's own bytes, every kernel beside it."""
    traffic = load_profiles_json("traffic.json")
    stats = load_profiles_json("kernel_stats.json")
    # committed PMC / rocprofv3 summaries are quoted only while they belong to the kernels being measured
    src = kernel_source_sha16()
    profiles_current = traffic.get("_kernel_source_sha16") == src and stats.get("_kernel_source_sha16") == src
    if not profiles_current:
        traffic, stats = {}, {}
    kernel_bytes = dict(KERNEL_BYTES_ACTUAL) if kernel_bytes is None else dict(kernel_bytes)
    texels = n * n * tiles
    key = lambda k: f"{k}@{n}" + (f"x{tiles}" if tiles > 1 else "")
    kernels = {}
    for k, ms in zip(names, kern_ms):
        us = ms * 1e3
        own = kernel_bytes.get(k, 0) * texels
        ent = {"launch_us": us, "own_bytes_per_launch": own, "achieved_GBps": own / (us * 1e-6) * 1e-9}
        ent["frac"] = ent["achieved_GBps"] / HBM_PEAK_GBPS
        tr = traffic.get(key(k))
        if tr:
            ent["traffic_bytes_per_launch"] = tr["hbm_bytes_per_launch"]
            ent["traffic_GBps"] = tr["hbm_bytes_per_launch"] / (us * 1e-6) * 1e-9
        st = stats.get(key(k))
        if st:
            ent["rocprof_launch_us"] = st["avg_us"]
            ent["event_over_rocprof"] = us / st["avg_us"]
        kernels[k] = ent
    # the timed (pipelined) regime's own counter passes: same counters, frames at the bench's pipeline depth (tools/pmc_depth.sh)
    pipelined_traffic = None
    if depth > 1:
        per = {k: traffic.get(key(k) + f":depth{depth}") for k in names}
        if all(per.values()):
            total = sum(v["hbm_bytes_per_launch"] for v in per.values())
            own_frame = own_bytes_per_texel * texels
            pipelined_traffic = {
                "bytes_per_launch": {k: v["hbm_bytes_per_launch"] for k, v in per.items()},
                "bytes_per_frame": total, "algorithmic_bytes_per_frame": own_frame, "over_algorithmic": total / own_frame,
                "counter_GBps": total / (ms_per_step * 1e-3) * 1e-9, "algorithmic_GBps": own_frame / (ms_per_step * 1e-3) * 1e-9,
                "source": per[names[0]]["source"],
                "what": f"FETCH_SIZE x 2 + WRITE_SIZE per launch with frames at pipeline depth {depth} (the store policies and cache state of the "
                        "timed region; the counter collection serialises the kernels), over this run's ms_per_step.  These counters sit at the "
                        "L2's memory-side port and count Infinity-Cache hits too (MI355X_MICROARCH.md, HBM section): they show that the "
                        "pipelined frame moves its algorithmic bytes and no more across that port -- how many of them the 256 MiB cache "
                        "serves is not observable with them"}
    dom = max(names, key=lambda k: kernels[k]["launch_us"])
    d = kernels[dom]
    tr = traffic.get(key(dom))
    frame_own = own_bytes_per_texel * texels
    return {
        "bound": "hbm", "kernel": dom,
        "achieved": d["achieved_GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": d["frac"],
        "traffic": tr["hbm_bytes_per_launch"] if tr else None,
        "traffic_source": tr.get("source") if tr else None,
        "committed_profiles": {"kernel_source_sha16": src, "current": profiles_current,
                               "note": "traffic / rocprof_launch_us come from profiles/traffic.json and profiles/kernel_stats.json (rocprofv3 runs "
                                       "of these kernels on another box, tools/prof_round.sh); they are quoted only while the hash of the kernel "
                                       "sources they were measured with equals the one being run, otherwise null"},
        "kernel_bytes_per_texel": kernel_bytes,
        "algorithmic_bytes_per_launch": d["own_bytes_per_launch"],
        "bytes_model": "this pipeline's own algorithmic bytes per texel (ocean_algorithmic_bytes_per_launch; fp32 seven-field frame:) k_zpass 23 (h0 8 + 16-bit dispersion of HALF the columns 1 in -- a column "
                       "and its point mirror share it --, half-size "
                       "intermediates 14 out), k_xpass_b 28 (10 in, raw height 2 + normal map 16 out), k_xpass_disp 22 (6 in, "
                       "displacement map 16 out); 73 per frame (DESIGN.md section 5)",
        "launch_us": d["launch_us"],
        "launch_us_source": "hipExtLaunchKernelGGL start/stop events on the launch stream (kernel execution time), serial frames, "
                            "mean over 200 frames of a pass of its own (independent of --steps / --warmup); rocprofv3 --kernel-trace "
                            "--stats of the same command: profiles/kernel_stats.json",
        "rocprof_launch_us": d.get("rocprof_launch_us"),
        "rocprof_frac": (d["own_bytes_per_launch"] / (d["rocprof_launch_us"] * 1e-6) * 1e-9 / HBM_PEAK_GBPS) if d.get("rocprof_launch_us") else None,
        "profiles_current": profiles_current,
        "regime": "serial frames (pipeline depth 1): every kernel has the GPU to itself",
        "kernels": kernels,
        "serial_us_per_step": serial_us_per_step,
        "serial_frame_GBps": frame_own / (serial_us_per_step * 1e-6) * 1e-9,
        "serial_frame_frac": frame_own / (serial_us_per_step * 1e-6) * 1e-9 / HBM_PEAK_GBPS,
        "pipelined_kernel_us": ({k: v * 1e3 for k, v in zip(names, kern_ms_pipe)} if kern_ms_pipe else None),
        "pipelined_traffic": pipelined_traffic,
        "pipelined_depth": depth,
        "frame_bytes_per_texel": own_bytes_per_texel,
        "frame_GBps": frame_own / (ms_per_step * 1e-3) * 1e-9,
        "frame_frac": frame_own / (ms_per_step * 1e-3) * 1e-9 / HBM_PEAK_GBPS,
        "frame_note": "frame_* = the timed (pipelined) region on this pipeline's 73 B/texel; part of that traffic is served by the "
                      "256 MiB Infinity Cache, so it is a rate of algorithmic bytes, bounded by what that mix can reach (DESIGN.md section 6)",
        "survey_model": {"what": "SURVEY.md 8d MODEL of a plain 3.5-transform two-pass scheme (108 B/texel; 40 / 40 / 28 per launch): "
                                 "model bytes divided by measured time -- NOT this pipeline's traffic, may exceed any physical rate",
                         "frame_bytes_per_texel": FRAME_BYTES_SURVEY,
                         "frame_model_GBps": FRAME_BYTES_SURVEY * texels / (ms_per_step * 1e-3) * 1e-9,
                         "dominant_kernel_model_GBps": KERNEL_BYTES_SURVEY[dom] * texels / (d["launch_us"] * 1e-6) * 1e-9},
    }


def _sig(x, digits=6):
    """Floats of the compact line carry `digits` significant digits (a 17-digit repr per number is what made the line outgrow the driver)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None                                   # strict JSON: no NaN / Infinity tokens
    return float(f"{x:.{digits}g}")


def _compact(obj, digits=6):
    if isinstance(obj, dict):
        return {k: _compact(v, digits) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_compact(v, digits) for v in obj]
    return _sig(obj, digits)


LINE_BYTES_MAX = 8192      # the final stdout line stays below this (the driver parsed 16.4 KB in round 3 and not 20.4 KB in round 4)
SIDECAR = "bench_extra.json"

ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "launch_us", "rocprof_launch_us",
                 "rocprof_frac", "algorithmic_bytes_per_launch", "frame_frac", "serial_frame_frac", "serial_us_per_step", "profiles_current")


def compact_roofline(r):
    """The roofline object of the stdout line: SURVEY.md 8d / the contract's keys for the dominant kernel, every kernel's duration and
    fraction beside them, no prose (the byte model, the sources of every figure and the pipelined-regime counters are in the sidecar)."""
    if not r:
        return None
    out = {k: r.get(k) for k in ROOFLINE_KEYS}
    out["kernels"] = {k: {"launch_us": v["launch_us"], "frac": v["frac"], "rocprof_launch_us": v.get("rocprof_launch_us"),
                          "bytes_per_texel": r["kernel_bytes_per_texel"].get(k), "traffic": v.get("traffic_bytes_per_launch")}
                      for k, v in r["kernels"].items()}
    return out


def compact_cpu(c, keys=("value", "unit", "cores", "nproc", "kind", "fft", "sample")):
    if not c:
        return None
    out = {k: c[k] for k in keys if k in c}
    if "sample" in out:
        out["sample"] = str(out["sample"])[:160]
    return out


def compact_gather(g):
    if not g:
        return None
    if "error" in g:
        return {"error": str(g["error"])[:200]}
    out = {k: g.get(k) for k in ("ranks", "rccl_ranks_seen", "tile_size", "tiles_per_rank", "bytes_into_root_per_step", "root_copy_matches_local_maps")}
    for k in ("compute_only", "compute_plus_gather_serial", "compute_gather_overlapped", "compute_gather_overlapped_half_maps"):
        if isinstance(g.get(k), dict):
            out[k + "_tiles_per_s"] = g[k].get("tiles_per_s")
    return out


def build_line(args, world, n, tiles, frames_per_s, ms_per_step, roofline, gather_obj, cpu, cpu_strong, timing=None, build_id=None):
    """The ONE JSON line of stdout (a dict): the contract's keys + compact roofline / cpu_baseline / gather / timing objects -- numbers
    and short labels only, < LINE_BYTES_MAX bytes.  Everything else this script measures goes to the sidecar (build_sidecar)."""
    t = timing or {}
    line = {
        "metric": "ocean frames/s (ComputeWaves, 7 fields -> displacement + normal map)",
        "value": frames_per_s, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"{tiles * world} x {n}x{n} tiles, {tiles} per GPU" if tiles * world > 1 else f"{n}x{n} tile") +
                               ", FULL7, reference default parameters" +
                               (" = BASELINE config 5" if (n, tiles * world, world) == (1024, 64, 8) else ""),
                   "tile_size": n, "tiles_per_rank": tiles, "pipeline_depth": args.depth, "seed": SEED, "dt": DT,
                   "parallelism": f"tiles sharded 1 process per GPU x{world}, no data-path collective"},
        "gtexels_per_s": n * n * frames_per_s * 1e-9,
        "prewarm_frames": args.prewarm,
        "timing": {"statistic": "median of regions", "regions": t.get("regions"), "ms_per_step_median": t.get("ms_per_step_median"),
                   "ms_per_step_mean": t.get("ms_per_step_mean"), "ms_per_step_p10": t.get("ms_per_step_p10"),
                   "ms_per_step_p90": t.get("ms_per_step_p90")} if t else None,
        "roofline": compact_roofline(roofline),
        "cpu_baseline": compact_cpu(cpu),
        "cpu_baseline_strong": compact_cpu(cpu_strong, keys=("value", "unit", "cores", "kind")),
        "gather": compact_gather(gather_obj),
        "kernel_source_sha16": kernel_source_sha16(),
        "build_id": build_id,
        "sidecar": SIDECAR,
    }
    return _compact(line)


def build_sidecar(line, args, roofline, gather_obj, extra_obj, cpu, cpu_strong, timing):
    """Everything measured beside the headline, with the prose that explains each figure: written to SIDECAR next to this script and
    echoed on stderr -- never part of the stdout line."""
    return {
        "line": line,
        "warmup_note": "`warmup` is the W of the command line; `prewarm_frames` more untimed frames run ahead of them (the device needs a "
                       "few hundred frames to reach its steady state); the roofline's per-kernel passes are separate runs of 200 frames "
                       "each, independent of --steps / --warmup",
        "warmup_frames_effective": args.prewarm + args.warmup,
        "api": "ocean_compute_waves_async x steps, then ocean_synchronize (frames alternate between "
               f"{args.depth} independent chains, each with its own intermediates and map set)",
        "timing": timing,
        "roofline": roofline,
        "cpu_baseline": cpu,
        "cpu_baseline_strong": cpu_strong,
        "speedup_vs_cpu_baseline_strong": (line["value"] / cpu_strong["value"]) if cpu_strong and cpu_strong.get("value") else None,
        "gather": gather_obj,
        "extra": extra_obj,
    }


def encode_line(line) -> bytes:
    """Strict JSON (no NaN / Infinity), one line, below LINE_BYTES_MAX: anything else is a bug of this script, not something to emit."""
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if "\n" in text or len(text) >= LINE_BYTES_MAX:
        raise RuntimeError(f"bench.py: the stdout line has {len(text)} bytes (limit {LINE_BYTES_MAX})")
    return (text + "\n").encode()


def measure_gather(W, torch, dist, wdist, dev, local_rank, world, rank, backend, red_dev, barrier):
    """SURVEY.md 8e's three figures on BASELINE config 5's per-GPU share (8 tiles of 1024^2 per rank): compute
    only, every batch followed by its gather, and the gather overlapped with the next batch (two map sets).  The
    gather is the library's own ocean_gather_maps (ncclGather x 2 on RCCL, zero-copy from the map buffers)."""
    n, tiles, reps = 1024, 8, 30
    b = W.OceanBatch(n, tiles, local_rank)
    first_tile, _ = wdist.tile_shard(tiles * world, world, rank)
    b.prepare(SEED + first_tile)
    uid = wdist.exchange_unique_id(W, src=0)
    b.comm_init(world, rank, uid)
    rccl_ranks, rccl_rank = b.comm_count()         # what the communicator itself says (ncclCommCount / ncclCommUserRank)
    recv = None
    if rank == 0:
        recv = torch.empty((2, world, tiles, n, n, 4), dtype=torch.float32, device=dev)
    rp = (recv[0].data_ptr(), recv[1].data_ptr()) if rank == 0 else (None, None)

    recv16 = torch.empty((2, world, tiles, n, n, 4), dtype=torch.float16, device=dev) if rank == 0 else None
    rp16 = (recv16[0].data_ptr(), recv16[1].data_ptr()) if rank == 0 else (None, None)

    def timed(depth, with_gather, sync_each, half=False):
        b.set_pipeline_depth(depth)
        for j in range(5):                                   # warm-up (also first touch of every chain's buffers)
            b.compute_waves_async(DT * j)
            if with_gather:
                b.gather_maps(0, *(rp16 if half else rp), half=half)
        b.synchronize(); barrier()
        t0 = time.perf_counter()
        for j in range(reps):
            b.compute_waves_async(DT * j)
            if with_gather:
                b.gather_maps(0, *(rp16 if half else rp), half=half)
            if sync_each:
                b.synchronize()
        b.synchronize(); barrier()
        return wdist.max_over_ranks((time.perf_counter() - t0) / reps, device=red_dev)

    compute = timed(2, False, False)
    serial = timed(1, True, True)
    overlapped = timed(2, True, False)
    overlapped16 = timed(2, True, False, half=True)
    # one more gathered frame, checked: the gather is a collective, so EVERY rank takes part; the root compares its own
    # tiles in the receive buffer with its maps and one tile of the last rank with that rank's checksum
    b.set_pipeline_depth(1)
    b.compute_waves_async(1.0); b.gather_maps(0, *rp); b.synchronize()
    import numpy as np
    d, q = b.read_maps(tiles - 1, 1)
    mine = torch.tensor([float(np.float64(d).sum()), float(np.float64(q).sum())], dtype=torch.float64, device=red_dev)
    sums = [torch.zeros_like(mine) for _ in range(world)] if world > 1 else [mine]
    if world > 1:
        dist.all_gather(sums, mine)
    ok = None
    if rank == 0:
        d0, q0 = b.read_maps(0, 1)
        ok = bool(np.array_equal(recv[0, 0, 0].cpu().numpy(), d0[0]) and np.array_equal(recv[1, 0, 0].cpu().numpy(), q0[0]))
        last = world - 1
        got = (float(recv[0, last, tiles - 1].double().sum()), float(recv[1, last, tiles - 1].double().sum()))
        want = (float(sums[last][0]), float(sums[last][1]))
        ok = ok and all(abs(g - w) <= 1e-6 * max(1.0, abs(w)) for g, w in zip(got, want))
    b.comm_destroy()
    b.close()
    return gather_report(n, tiles, world, rccl_ranks, rccl_rank, compute, serial, overlapped, overlapped16, ok)


def gather_report(n, tiles, world, rccl_ranks, rccl_rank, compute_s, serial_s, overlapped_s, overlapped16_s, ok):
    """measure_gather's result as a dictionary (pure: the seconds per step of the four regimes in, SURVEY.md 8e's figures out).  Kept apart
    from the measurement so that the CPU contract test builds the N > 1 line from exactly what a real run would hand to build_line
    (tests/line_schema.py checks both)."""
    per_rank = tiles * n * n * 4 * 4 * 2
    total = world * tiles
    return {"what": "BASELINE config 5 share: 8 tiles of 1024x1024 per rank per step; ocean_gather_maps = ncclGather x 2 "
                    "(displacement, normal) in one RCCL group from the library's map buffers to rank 0; 'overlapped' = depth 2, "
                    "the gather of batch j runs on the communication stream beside the synthesis of batch j+1",
            "transport": "RCCL (librccl loaded by libocean_hip.so), ncclGather" + ("; 1 rank: device-local copy, no xGMI traffic" if world == 1 else ""),
            "tile_size": n, "tiles_per_rank": tiles, "ranks": world, "rccl_ranks_seen": rccl_ranks, "rccl_rank_of_root": rccl_rank,
            "bytes_per_rank_per_step": per_rank,
            "bytes_into_root_per_step": per_rank * (world - 1),
            "compute_only": {"ms_per_step": compute_s * 1e3, "tiles_per_s": total / compute_s},
            "compute_plus_gather_serial": {"ms_per_step": serial_s * 1e3, "tiles_per_s": total / serial_s},
            "compute_gather_overlapped": {"ms_per_step": overlapped_s * 1e3, "tiles_per_s": total / overlapped_s},
            "compute_gather_overlapped_half_maps": {"ms_per_step": overlapped16_s * 1e3, "tiles_per_s": total / overlapped16_s,
                                                    "what": "ocean_gather_maps_f16: maps converted to IEEE half on the sender, 16 B/texel on the wire"},
            "root_ingest_GBps_overlapped": per_rank * (world - 1) / overlapped_s * 1e-9,
            "root_copy_matches_local_maps": ok}


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.regions < 7:
        raise SystemExit("--regions must be >= 7 (ms_per_step is the median region, p10 / p90 beside it)")
    if args.cpu_baseline_child:                     # no GPU, no torch: just the oracle on the host cores
        ref, strong = cpu_baseline(args.size, args.cpu_seconds)
        check = None
        if args.check_file:
            try:
                check = check_frames_against_oracle(args.check_file)
            except Exception as exc:                # the check is an extra: its failure is reported, the baseline stands
                check = {"error": f"{type(exc).__name__}: {exc}"}
        print(json.dumps({"cpu_baseline": ref, "cpu_baseline_strong": strong, "check": check}))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))          # nothing in this process has touched a GPU
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")     # CPU baseline: idle OpenMP workers must not spin beside the library FFT's threads
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a different n_gpus")

    # stdout carries exactly ONE line, the JSON: everything else that writes to file descriptor 1 (RCCL prints a
    # version banner there when a communicator is created) is sent to stderr until the line is emitted
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # OCEAN_BENCH_BACKEND=gloo is a developer switch: it lets the multi-rank control flow be exercised on a
    # box with fewer GPUs than ranks (ranks then share devices; RCCL refuses that).  The driver never sets it.
    backend = os.environ.get("OCEAN_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = str(dev) if backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import watersurfacerendering_amd as W
    from watersurfacerendering_amd import dist as wdist
    build_id = W._abi.lib().ocean_build_id().decode()      # content hash of the sources the loaded library was built from

    n, tiles = args.size, args.tiles
    first_tile, _ = wdist.tile_shard(tiles * world, world, rank)
    b = W.OceanBatch(n, tiles, local_rank)
    b.set_pipeline_depth(args.depth)
    b.prepare(SEED + first_tile)

    def barrier():
        if world > 1:
            dist.barrier()

    def sync():
        b.synchronize()
        torch.cuda.synchronize()

    serial_passes = []
    nk = 200                    # launches averaged per kernel, whatever --steps is (a 20-step run would average 20 noisy ones)

    def serial_pass(when):
        """Per-kernel durations of strictly serial frames in a context of its own (what a caller of the synchronous ComputeWaves has): kernel
        execution times from events attached to the dispatches, every kernel with the GPU to itself."""
        bs = W.OceanBatch(n, tiles, local_rank)
        bs.prepare(SEED + first_tile)
        # The device comes out of an idle gap -- context creation is one -- at 1.55-1.7 GHz and needs ~25 ms of load to reach its sustained
        # 2.26-2.31 GHz (in-kernel clock, profiles/r06_slow_window.txt); the z pass is bound by shader cycles (45.5-46.0 k per launch at any
        # clock), so 300 warm-up frames (17 ms at 2048^2) left part of the timed ones inside that ramp: rounds 4-5's "slow window".  The first 100
        # frames are timed as they come (state "clock ramping", reported in the sidecar), then 60 ms of frames bring the clock up.
        _, kern_cold = bs.time_frames(0.0, DT, 0, 100, per_kernel=True)
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < 0.06:
            for j in range(50):
                bs.compute_waves_async(DT * j)
            bs.synchronize()
        ms_i, kern_i = bs.time_frames(0.0, DT, 100, nk, per_kernel=True)
        placed = bs.placement_report()
        bs.close()
        serial_passes.append((ms_i / nk * 1e3, kern_i, when, kern_cold, placed))
    for j in range(min(args.prewarm, 200)):            # (the first frames of the process: module load, first touch)
        b.compute_waves_async(DT * j)
    sync()
    serial_pass("before the timed regions")
    for j in range(args.prewarm):                      # device pre-warm (clocks, caches, first touch of every chain's buffers)
        b.compute_waves_async(DT * j)
    sync()
    for j in range(args.warmup):
        b.compute_waves_async(DT * j)
    # K timed regions of exactly --steps steps each, every one bracketed by barrier + synchronise on both sides and reduced to the MAX
    # over ranks; ms_per_step / value come from the MEDIAN region, the spread (p10, p90, min, max) is reported beside it.
    region_s = []
    for r in range(args.regions):
        sync(); barrier(); sync()
        t0 = time.perf_counter()
        for j in range(args.steps):
            b.compute_waves_async(DT * (args.warmup + r * args.steps + j))
        sync()
        t1 = time.perf_counter()         # this rank's K steps are done; the closing barrier + synchronise follow, and the region's time is
        barrier(); sync()                # the MAX over ranks of (t1 - t0): all ranks left the opening barrier together, so that is the time
        region_s.append(wdist.max_over_ranks(t1 - t0, device=red_dev))     # until the slowest one finished -- without the barrier's own latency
    import numpy as _np
    reg = _np.sort(_np.asarray(region_s, dtype=_np.float64))
    elapsed = float(_np.median(reg))
    ms_per_step = elapsed / args.steps * 1e3
    frames_per_s = world * tiles * args.steps / elapsed
    timing_stats = {"regions": args.regions, "steps_per_region": args.steps, "statistic": "median over the regions",
                    "ms_per_step_median": ms_per_step,
                    "ms_per_step_mean": float(reg.mean()) / args.steps * 1e3,
                    "ms_per_step_p10": float(_np.percentile(reg, 10)) / args.steps * 1e3,
                    "ms_per_step_p90": float(_np.percentile(reg, 90)) / args.steps * 1e3,
                    "ms_per_step_min": float(reg[0]) / args.steps * 1e3, "ms_per_step_max": float(reg[-1]) / args.steps * 1e3,
                    "ms_per_step_of_each_region_in_order": [x / args.steps * 1e3 for x in region_s],
                    "what": "each region = exactly `steps` asynchronous frames between barrier + ocean_synchronize + torch.cuda.synchronize "
                            "on both sides, max over ranks; regions run back to back after the warm-up"}

    # ---- per-kernel durations (kernel execution time from events attached to the dispatches), measured with
    # serial frames -- a per-launch duration only characterises a kernel that has the GPU to itself; the
    # pipelined durations are reported beside them
    kern_ms_pipe = None
    kern_ms_main = None
    if args.depth > 1:
        _, kern_ms_pipe = b.time_frames(0.0, DT, 200, nk, per_kernel=True)
        b.set_pipeline_depth(1)
        _, kern_ms_main = b.time_frames(0.0, DT, 200, nk, per_kernel=True)      # serial frames of THIS context, reported beside
    names = b.kernel_names()
    own_bpt = float(b.algorithmic_bytes_per_texel)
    kernel_bytes = dict(zip(names, b.algorithmic_bytes_per_launch()))
    b.close()
    torch.cuda.empty_cache()
    # The serial pass runs in a context of its own at depth 1 -- what a caller of the synchronous ComputeWaves has -- and the
    # serial frames of the bench context are reported beside it (roofline.serial_kernels_in_bench_context_us).  The hardware queue a
    # context's stream lands on moves these figures by +-1 us per kernel between passes (profiles/r03_bimodal_probe.txt 4a); the slow
    # k_xpass_b of earlier rounds (one process in 4...25) is gone since its victim workgroup is dispatched first (profiles/r03_xpass_trace.txt).
    # THREE such passes, spread over the run -- before the timed regions (above), right behind them (here), and at the very end (rank 0, behind
    # the secondary measurements) -- and the one with the median frame time is reported; all three are in the sidecar.  Reason: now and then
    # the serial z pass of a 2048^2 frame runs 15-25 % slower for a WINDOW of a process's life (every context created in that window, e.g. all
    # the contexts measured right behind the timed regions in one run, none in the next; the x passes beside it unaffected) -- not the clocks
    # (tools/clock_ramp.py), not the code or the buffers' placement (tools/ctx_spread.py, tools/ctx_churn.py: dozens of contexts, one value);
    # unexplained.  A single pass reports whichever window it falls into.
    serial_pass("right behind the timed regions")

    def make_roofline():
        # (the median of an odd number of passes; of an even number -- the line the gather watchdog emits before the last pass -- the SLOWER
        #  middle one: never the optimistic pass under a 'median' label)
        frame_us, kern = sorted(serial_passes, key=lambda p: p[0])[len(serial_passes) // 2][:2]
        r = roofline_object(n, tiles, names, kern, kern_ms_pipe, args.depth, ms_per_step, frame_us, own_bpt, kernel_bytes)
        r["serial_pass"] = ("contexts of their own at pipeline depth 1 (the synchronous-call configuration), 300 warm-up + 200 timed frames each; one before "
                            "the timed regions, one right behind them, one at the end of the run (rank 0): the one with the median frame time is reported")
        r["serial_passes_used"] = len(serial_passes)
        r["z_pass_states_us"] = {"what": "the dominant kernel in the two states of the device's shader clock: the first 100 serial frames of a fresh context "
                                         "(right after an idle gap: the clock ramps from ~1.6 GHz) and frames behind 60 ms of load (sustained clock: what "
                                         "roofline.launch_us reports); per pass, in the order of serial_passes_us",
                                 "clock_ramping": [p[3][0] * 1e3 for p in serial_passes], "sustained_clock": [p[1][0] * 1e3 for p in serial_passes]}
        r["serial_passes_us"] = [{"when": p[2], "frame": p[0], **{k: v * 1e3 for k, v in zip(names, p[1])}} for p in serial_passes]
        r["placement_search"] = {"what": "ocean_prepare's placement search in each serial pass's context (include/ocean_dev.h): candidate allocations of spectrum + "
                                         "intermediates timed, serial frame us of the one kept and of the slowest -- the spread is what a context without the "
                                         "search could have drawn", "per_pass": [{"candidates": p[4][0], "us_chosen": p[4][1], "us_slowest": p[4][2]} for p in serial_passes]}
        if kern_ms_main is not None:
            r["serial_kernels_in_bench_context_us"] = {k: v * 1e3 for k, v in zip(names, kern_ms_main)}
        return r
    roofline = make_roofline()

    # ---- the exchange step: RCCL gather of the packed maps, outside the timed region
    gather = None
    emitted = [False]

    def emit(line_obj, sidecar_obj=None):
        """rank 0: the sidecar (file beside this script + stderr), then the ONE JSON line through the saved stdout descriptor; at most once"""
        if rank != 0 or emitted[0]:
            return
        emitted[0] = True
        import ctypes
        data = encode_line(line_obj)
        if sidecar_obj is not None:
            text = json.dumps(sidecar_obj, indent=1, default=str)
            try:
                with open(os.path.join(ROOT, SIDECAR), "w") as f:
                    f.write(text + "\n")
            except OSError as exc:                    # a read-only checkout: stderr still carries it
                print(f"bench.py: could not write {SIDECAR}: {exc}", file=sys.stderr)
            print("bench.py sidecar (" + SIDECAR + "):\n" + text, file=sys.stderr)
        sys.stdout.flush(); sys.stderr.flush()
        ctypes.CDLL(None).fflush(None)            # C stdio buffers (the RCCL banner) leave through the redirected descriptor
        os.write(json_fd, data)

    def headline(gather_obj, extra_obj, cpu, cpu_strong):
        line = build_line(args, world, n, tiles, frames_per_s, ms_per_step, roofline, gather_obj, cpu, cpu_strong, timing_stats, build_id)
        return line, build_sidecar(line, args, roofline, gather_obj, extra_obj, cpu, cpu_strong, timing_stats)

    # Everything the contract asks for is measured by now.  What follows at N > 1 -- the gather over xGMI, which no machine
    # available to the builder could run -- must not be able to take the line down with it: if it has not come back after
    # four minutes, rank 0 emits the line with the failure recorded under "gather" and every rank leaves.
    watchdog = None
    if world > 1:
        import threading

        def give_up():
            # the line goes out (the timed region is complete), but the run FAILED: launcher and driver must see it
            emit(*headline({"error": "the gather measurement did not finish within 240 s; timed region unaffected"}, {}, None, None))
            os._exit(3)
        watchdog = threading.Timer(240.0, give_up)
        watchdog.daemon = True
        watchdog.start()
    if not args.no_gather and (backend == "nccl" or world == 1):
        try:
            gather = measure_gather(W, torch, dist, wdist, dev, local_rank, world, rank, backend, red_dev, barrier)
        except Exception as exc:                    # a failed exchange step is reported, it does not void the timed region
            if world == 1:
                raise
            gather = {"error": f"{type(exc).__name__}: {exc}"}
    if watchdog is not None:
        watchdog.cancel()

    out = (None, None)
    if rank == 0:
        extra = {}
        if not args.no_extra and world == 1:
            # strictly serial frames (what a caller of the synchronous ComputeWaves sees, minus the read-back)
            extra["2048x2048_serial_frames_depth1"] = measure_config(W, 2048, 1, local_rank, 1000, 300, depth=1)
            # BASELINE.json configs beside the headline one (parity for all of them: tests/)
            extra["512x512_single_tile_depth1"] = measure_config(W, 512, 1, local_rank, 2000, 500)
            extra["512x512_single_tile_depth4"] = measure_config(W, 512, 1, local_rank, 2000, 500, depth=4)
            extra["512x512_choppy5_single_tile_depth1"] = measure_config(W, 512, 1, local_rank, 2000, 500, mode=1)
            extra["256x256_height1_single_tile_depth1"] = measure_config(W, 256, 1, local_rank, 2000, 500, mode=2)
            extra["dropin_call"] = measure_dropin_call(W)
            extra["512x512_synchronous_calls"] = measure_sync_calls(W, 512, local_rank)
            extra["2048x2048_synchronous_calls"] = measure_sync_calls(W, 2048, local_rank)
            extra["vertex_stage_512"] = measure_consumer(W, 512, local_rank)
            extra["vertex_stage_2048"] = measure_consumer(W, 2048, local_rank)
            extra["512x512_batch16"] = measure_config(W, 512, 16, local_rank, 1000, 300)
            extra["512x512_batch16_depth2"] = measure_config(W, 512, 16, local_rank, 1000, 300, depth=2)
            extra["1024x1024_batch8_per_gpu_share_of_config5"] = measure_config(W, 1024, 8, local_rank, 500, 150)
            extra["1024x1024_batch8_per_gpu_share_of_config5_depth2"] = measure_config(W, 1024, 8, local_rank, 500, 150, depth=2)
            # BASELINE config 4 as it is named -- "4096^2, fp32 vs fp16 spectrum, tolerance vs double precision stated" -- plus this
            # pipeline's own reduced-precision mode (half2 intermediates); each line's error against the float64 oracle is measured in
            # this run by the CPU-baseline child (`error_vs_float64_oracle`, filled in below)
            extra["4096x4096_fp32_depth3"] = measure_config(W, 4096, 1, local_rank, 300, 100, depth=3)
            extra["4096x4096_fp16_spectrum_depth3"] = measure_config(W, 4096, 1, local_rank, 300, 100, depth=3, h0_bits=16)
            # BASELINE config 4's reduced-precision mode: half2 intermediates between the passes (59 instead of 73 B/texel;
            # maps within 1e-3 of the fp32 path's, tests/test_parity_gpu.py) -- NOT the headline, which is fp32 throughout
            extra["4096x4096_fp16_intermediates_depth3"] = measure_config(W, 4096, 1, local_rank, 300, 100, depth=3, inter_bits=16)
            extra["2048x2048_fp16_intermediates_depth3"] = measure_config(W, 2048, 1, local_rank, 1000, 300, depth=3, inter_bits=16)
            extra["1024x1024_batch8_fp16_intermediates_depth2"] = measure_config(W, 1024, 8, local_rank, 500, 150, depth=2, inter_bits=16)
            # the Jacobian / foam channel (eight fields, 85 B/texel)
            extra["2048x2048_jacobian_depth3"] = measure_config(W, 2048, 1, local_rank, 1000, 300, depth=3, mode=3)
        cpu = cpu_strong = None
        if not args.no_cpu_baseline:                # rank 0's host cores, at every N (a child process: no GPU, no process group)
            check_file = None
            if "4096x4096_fp32_depth3" in extra:
                import tempfile
                tmpdir = tempfile.mkdtemp(prefix="ocean_bench_")
                try:
                    check_file = save_config4_frames(W, local_rank, os.path.join(tmpdir, "config4.npz"))
                except Exception as exc:
                    extra["config4_error_check"] = {"error": f"{type(exc).__name__}: {exc}"}
            cpu, cpu_strong, check = cpu_baseline_isolated(n, args.cpu_seconds, check_file)
            if check_file:
                import shutil
                shutil.rmtree(os.path.dirname(check_file), ignore_errors=True)
                extra["config4_error_check"] = check
                if check and "variants" in check:
                    for key, var in (("4096x4096_fp32_depth3", "fp32"), ("4096x4096_fp16_spectrum_depth3", "fp16_spectrum"),
                                     ("4096x4096_fp16_intermediates_depth3", "fp16_intermediates")):
                        if key in extra and var in check["variants"]:
                            extra[key]["error_vs_float64_oracle"] = check["variants"][var]["max_err_over_max_channel"]
                            extra[key]["stated_tolerance"] = 1e-5 if var == "fp32" else 1e-3
        serial_pass("at the end of the run")
        roofline = make_roofline()
        out = headline(gather, extra, cpu, cpu_strong)
    emit(*out)
    failed = isinstance(gather, dict) and "error" in gather
    if world > 1:
        import threading
        # the line is out: a stuck teardown must not hold the launcher -- but it is a failure, and reported as one
        t = threading.Timer(300.0, lambda: os._exit(4))     # (rank 0 arrives after its CPU baseline leg: up to ~2 minutes)
        t.daemon = True
        t.start()
        dist.barrier()
        dist.destroy_process_group()
        t.cancel()
    if failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
