#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X ocean synthesiser.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One *step* = one ComputeWaves(t) of the hot path (reference
WSTessendorf.cpp:284-455) on synthetic input: a 2048 x 2048 tile, all seven
output fields, reference default parameters, xi from the counter-based RNG
with seed 0x5EED0000 + tile_index, t_j = 0.05*j.  Inputs (h0,
omega) are resident in HBM before the timed region; outputs are the two finished
RGBA32F maps in HBM (D2H read-back is not part of the metric).

The K timed steps are K asynchronous ocean_compute_waves_async calls followed by one
synchronise.  With --depth 3 (default) consecutive frames rotate over three
independent chains (own stream, own intermediates, own map set), so one frame's first
pass overlaps the others' map passes; every frame is still computed in full and its
maps stay addressable until the chain is reused.  --depth 1 = strictly serial frames
(also reported under extra).  Per-launch durations for the roofline object come from HIP
events around every launch of serial frames (a second pass of the same frames).

Multi-GPU: tiles are independent, so every rank synthesises its own tile(s)
with no data-path collective ("weak" scaling, value = frames of all ranks per
second).  The north-star's single RCCL gather of the packed maps is measured
separately after the timed region and reported under "gather" (it is
xGMI-bound and slower than one GPU's synthesis: DESIGN.md section 6).

Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x5EED0000
DT = 0.05
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Bytes per texel per launch, two accountings:
#  * SURVEY.md 8d's model (7 fields, two passes, F = 3.5 complex intermediates, no point
#    symmetry): 108 B/texel per frame, apportioned to the launches that do that work;
#    `roofline.achieved` uses this one, as the task statement prescribes.
#  * what this pipeline actually has to move (half-size intermediates, 16-bit dispersion): 74 B/texel per frame.
KERNEL_BYTES_SURVEY = {"k_zpass": 40, "k_xpass_b": 40, "k_xpass_disp": 28}
KERNEL_BYTES_ACTUAL = {"k_zpass": 24, "k_xpass_b": 28, "k_xpass_disp": 22}
FRAME_BYTES_SURVEY = 108.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the device needs a few hundred frames (tens of ms) to reach its steady state -- 200 frames after 10
    # warm-up frames measure 60 us/frame, every later batch of 200 measures 53-54 -- and 3000 frames are 0.2 s
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--prewarm", type=int, default=500, help="untimed frames before the W warm-up steps (brings the device to its steady state even when W is small)")
    ap.add_argument("--size", type=int, default=2048, help="tile size N (default: the roofline config, 2048)")
    ap.add_argument("--tiles", type=int, default=1, help="independent tiles per rank per step")
    ap.add_argument("--depth", type=int, default=3, help="frame pipeline depth of the asynchronous API (1 = strictly serial frames)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for the CPU baseline sample")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL gather measurement at N>1")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary 512^2 / batched measurements")
    return ap.parse_args()


def cpu_baseline(n: int, budget_s: float):
    """The oracle (reference-shaped OpenMP port, own float FFT) timed on the host cores."""
    from oracle import oracle as O
    o = O.Oracle(n)
    o.prepare(seed=SEED)
    threads = int(O.lib().oracle_num_threads())
    for j in range(2):
        o.compute_waves(DT * j, fft=O.FFT_F32, copy=False)
    times = []
    t_start = time.perf_counter()
    j = 0
    while True:
        t0 = time.perf_counter()
        o.compute_waves(DT * (2 + j), fft=O.FFT_F32, copy=False)
        times.append(time.perf_counter() - t0)
        j += 1
        if (time.perf_counter() - t_start >= budget_s and j >= 5) or j >= 200:
            break
    times.sort()
    med = times[len(times) // 2]
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
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
    # for scale only: the FFT stage alone with a tuned library FFT (scipy's pocketfft, every core) --
    # what a faster CPU FFT than the port's would buy the reference; not the reported baseline
    pocket = None
    try:
        import numpy as np
        import scipy.fft
        x = (np.random.default_rng(0).standard_normal((7, n, n)) + 0j).astype(np.complex64)
        scipy.fft.ifft2(x, workers=os.cpu_count())
        tp = []
        for _ in range(3):
            t0 = time.perf_counter()
            scipy.fft.ifft2(x, workers=os.cpu_count())
            tp.append(time.perf_counter() - t0)
        pocket = {"ms": min(tp) * 1e3, "what": f"scipy pocketfft: seven complex64 {n}x{n} 2-D inverse FFTs, workers={os.cpu_count()} "
                                              "(FFT stage only: no spectrum animation, no pack, no normalisation)"}
    except Exception:
        pass
    return {
        "value": 1.0 / med, "unit": "frames/s", "cores": threads, "kind": "port",
        "fft_stage_with_library_fft": pocket,
        "sample": f"{len(times)} frames of the same {n}x{n} 7-field workload after 2 warm-up frames, median "
                  f"({med * 1e3:.1f} ms/frame); FFTW not available on this host: baseline is the oracle's own "
                  f"float Stockham FFT in the reference's OpenMP shape (7 single-threaded 2-D FFTs in parallel)",
        "gtexels_per_s": n * n / med * 1e-9,
        "host": {"cpu_model": model, "nproc": os.cpu_count(), "omp_max_threads": threads,
                 "note": "stage D of the reference shape (7 single-threaded 2-D FFTs in omp sections) cannot use more than 7 threads"},
        "config1_256x256_height_only_cpu_ms": t1[len(t1) // 2] * 1e3,
    }


def measure_config(W, n, tiles, device, steps, warmup, h0_bits=32, depth=1, mode=0):
    b = W.OceanBatch(n, tiles, device)
    if h0_bits != 32:
        b.set_spectrum_precision(h0_bits)
    if mode:
        b.set_mode(mode)
    b.set_pipeline_depth(depth)
    b.prepare(SEED)
    ms, kern = b.time_frames(0.0, DT, warmup, steps, per_kernel=True)
    per = ms / steps * 1e-3
    KERNEL_ORDER = b.kernel_names()
    b.close()
    fb = {0: FRAME_BYTES_SURVEY, 1: 92.0, 2: 44.0}[mode] - (4.0 if h0_bits == 16 else 0.0)   # SURVEY.md 8d per mode
    return {"size": n, "tiles_per_step": tiles, "pipeline_depth": depth, "mode": ["FULL7", "CHOPPY5", "HEIGHT1"][mode],
            "frames_per_s": tiles / per, "us_per_step": per * 1e6,
            "gtexels_per_s": n * n * tiles / per * 1e-9, "algorithmic_GBps": fb * n * n * tiles / per * 1e-9,
            "kernel_us": {k: v * 1e3 for k, v in zip(KERNEL_ORDER, kern)}}


def measure_sync_calls(W, n, device, calls=300):
    """Median latency of the synchronous ComputeWaves (the reference's call shape: returns the amplitude)."""
    import numpy as np
    b = W.OceanBatch(n, 1, device)
    b.prepare(SEED)
    for j in range(30):
        b.compute_waves(DT * j)
    ts = np.empty(calls)
    for j in range(calls):
        t0 = time.perf_counter()
        b.compute_waves(DT * j)
        ts[j] = time.perf_counter() - t0
    b.close()
    return {"size": n, "calls": calls, "median_us_per_call": float(np.median(ts) * 1e6), "p95_us_per_call": float(np.percentile(ts, 95) * 1e6),
            "what": "host-side wall time of ocean_compute_waves (enqueue + the frame + one stream synchronisation; min/max keys arrive in host-coherent memory), called from Python"}


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


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

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

    for j in range(args.prewarm):                      # device pre-warm (clocks, caches, first touch of every chain's buffers)
        b.compute_waves_async(DT * j)
    sync()
    for j in range(args.warmup):
        b.compute_waves_async(DT * j)
    sync(); barrier(); sync()
    t0 = time.perf_counter()
    for j in range(args.steps):
        b.compute_waves_async(DT * (args.warmup + j))
    sync(); barrier(); sync()
    elapsed = time.perf_counter() - t0
    elapsed = wdist.max_over_ranks(elapsed, device=red_dev)
    ms_per_step = elapsed / args.steps * 1e3
    frames_per_s = world * tiles * args.steps / elapsed

    # ---- dominant-kernel roofline, measured live with HIP events on the launch stream.
    # Per-launch durations only characterise a kernel when it has the GPU to itself, so the
    # roofline object is measured with serial frames (depth 1; `bench.py --depth 1` under
    # rocprofv3 reproduces them: profiles/).  At depth > 1 launches of consecutive frames
    # overlap; those durations are reported beside it, and the frame-level fractions cover
    # the pipelined regime.
    kern_ms_pipe = None
    if args.depth > 1:
        _, kern_ms_pipe = b.time_frames(0.0, DT, 50, min(args.steps, 200), per_kernel=True)
        b.set_pipeline_depth(1)
    ms_serial, kern_ms = b.time_frames(0.0, DT, 50, min(args.steps, 200), per_kernel=True)
    serial_us_per_step = ms_serial / min(args.steps, 200) * 1e3
    b.set_pipeline_depth(args.depth)
    KERNEL_ORDER = b.kernel_names()
    dom = max(range(3), key=lambda i: kern_ms[i])
    dom_name = KERNEL_ORDER[dom]
    dom_bytes = KERNEL_BYTES_SURVEY[dom_name] * n * n * tiles
    dom_bytes_actual = KERNEL_BYTES_ACTUAL[dom_name] * n * n * tiles
    achieved = dom_bytes / (kern_ms[dom] * 1e-3) * 1e-9
    traffic = None
    traffic_src = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            ent = tj.get(f"{dom_name}@{n}")
            if ent:
                traffic, traffic_src = ent["hbm_bytes_per_launch"], ent.get("source")
        except Exception:
            pass
    roofline = {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": dom_bytes, "launch_us": kern_ms[dom] * 1e3,
                "regime": "serial frames (pipeline depth 1): HIP events around every launch, one frame at a time; "
                          "event intervals include ~2.5-3 us of launch/event processing per launch vs rocprofv3",
                "serial_us_per_step": serial_us_per_step,
                "serial_frame_frac": FRAME_BYTES_SURVEY * n * n * tiles / (serial_us_per_step * 1e-6) * 1e-9 / HBM_PEAK_GBPS,
                "pipelined_kernel_us": ({k: v * 1e3 for k, v in zip(KERNEL_ORDER, kern_ms_pipe)} if kern_ms_pipe else None),
                "pipelined_depth": args.depth,
                "bytes_model": "SURVEY.md 8d (108 B/texel per frame) apportioned per launch",
                "achieved_on_this_pipelines_own_bytes": dom_bytes_actual / (kern_ms[dom] * 1e-3) * 1e-9,
                "own_bytes_per_launch": dom_bytes_actual,
                "kernel_us": {k: v * 1e3 for k, v in zip(KERNEL_ORDER, kern_ms)},
                "frame_bytes_per_texel_survey_8d": FRAME_BYTES_SURVEY,
                "frame_algorithmic_GBps": FRAME_BYTES_SURVEY * n * n * tiles / (ms_per_step * 1e-3) * 1e-9,
                "frame_frac": FRAME_BYTES_SURVEY * n * n * tiles / (ms_per_step * 1e-3) * 1e-9 / HBM_PEAK_GBPS}

    # ---- RCCL gather of the packed maps (north-star exchange step), outside the timed region
    gather = None
    if world > 1 and not args.no_gather:
        reps = 10
        # layout [2 (displacement, normal)][tiles][N][N][4] = the library's own tile-major map arrays;
        # binding caller-owned output makes the frames strictly serial (one map set)
        maps = torch.empty((2, tiles, n, n, 4), dtype=torch.float32, device=dev)
        b.bind_output(maps[0].data_ptr(), maps[1].data_ptr())
        sync(); barrier()
        tg = time.perf_counter()
        for j in range(reps):
            b.compute_waves_async(DT * j)
            b.synchronize()
            wdist.gather_maps(maps if backend == "nccl" else maps.cpu(), dst=0)
            torch.cuda.synchronize()          # the collective runs on its own stream: finish it before `maps` is rewritten
        barrier()
        serial = wdist.max_over_ranks((time.perf_counter() - tg) / reps, device=red_dev)
        # the same with the gather of frame j-1 overlapped with the synthesis of frame j (two map sets,
        # the collective on its own stream): SURVEY.md 8e's third figure
        maps2 = [maps, torch.empty_like(maps)]
        sync(); barrier()
        tg = time.perf_counter()
        work = None
        for j in range(reps + 1):
            if j < reps:
                cur = maps2[j % 2]
                b.bind_output(cur[0].data_ptr(), cur[1].data_ptr())
                b.compute_waves_async(DT * j)
            if j > 0:
                prev = maps2[(j - 1) % 2]
                _, work = wdist.gather_maps(prev if backend == "nccl" else prev.cpu(), dst=0, async_op=True)
            b.synchronize()
            if work is not None:
                work.wait()
                torch.cuda.synchronize()      # host-side completion: the next frame reuses that map set
        barrier()
        overlapped = wdist.max_over_ranks((time.perf_counter() - tg) / reps, device=red_dev)
        gather = {"what": "every step followed by one torch.distributed.gather (RCCL) of the packed maps to rank 0; "
                          "'overlapped' = the gather of frame j-1 runs beside the synthesis of frame j (two map sets)",
                  "backend": backend, "bytes_per_rank": int(maps.numel() * 4), "ms_per_step_serial": serial * 1e3,
                  "frames_per_s_serial": world * tiles / serial, "ms_per_step_overlapped": overlapped * 1e3,
                  "frames_per_s_overlapped": world * tiles / overlapped}

    out = None
    if rank == 0:
        extra = {}
        if not args.no_extra and world == 1:
            b.close()
            torch.cuda.empty_cache()
            # strictly serial frames (what a caller of the synchronous ComputeWaves sees, minus the read-back)
            extra["2048x2048_serial_frames_depth1"] = measure_config(W, n, tiles, local_rank, 1000, 300, depth=1)
            # BASELINE.json configs beside the headline one (parity for all of them: tests/test_parity_gpu.py)
            extra["512x512_single_tile_depth1"] = measure_config(W, 512, 1, local_rank, 2000, 500)
            extra["512x512_single_tile_depth4"] = measure_config(W, 512, 1, local_rank, 2000, 500, depth=4)
            extra["512x512_choppy5_single_tile_depth1"] = measure_config(W, 512, 1, local_rank, 2000, 500, mode=1)
            extra["256x256_height1_single_tile_depth1"] = measure_config(W, 256, 1, local_rank, 2000, 500, mode=2)
            extra["512x512_synchronous_calls"] = measure_sync_calls(W, 512, local_rank)
            extra["2048x2048_synchronous_calls"] = measure_sync_calls(W, 2048, local_rank)
            extra["vertex_stage_512"] = measure_consumer(W, 512, local_rank)
            extra["vertex_stage_2048"] = measure_consumer(W, 2048, local_rank)
            extra["512x512_batch16"] = measure_config(W, 512, 16, local_rank, 1000, 300)
            extra["512x512_batch16_depth2"] = measure_config(W, 512, 16, local_rank, 1000, 300, depth=2)
            extra["1024x1024_batch8_per_gpu_share_of_config5"] = measure_config(W, 1024, 8, local_rank, 500, 150)
            extra["1024x1024_batch8_per_gpu_share_of_config5_depth2"] = measure_config(W, 1024, 8, local_rank, 500, 150, depth=2)
            extra["4096x4096_fp32_spectrum_depth3"] = measure_config(W, 4096, 1, local_rank, 300, 100, depth=3)
            extra["4096x4096_fp16_spectrum_depth3"] = measure_config(W, 4096, 1, local_rank, 300, 100, h0_bits=16, depth=3)
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(n, args.cpu_seconds)
        out = {
            "metric": "ocean frames/s (ComputeWaves, 7 fields -> displacement + normal map)",
            "value": frames_per_s, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "prewarm_frames": args.prewarm, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n}x{n} tile, FULL7 (7 real fields via 3.5 complex 2-D iFFTs), "
                                   f"{tiles} tile(s) per rank per step, reference default parameters",
                       "tile_size": n, "tiles_per_rank": tiles, "seed": SEED, "dt": DT,
                       "pipeline_depth": args.depth,
                       "api": "ocean_compute_waves_async x steps, then ocean_synchronize (frames alternate between "
                              f"{args.depth} independent chains, each with its own intermediates and map set)",
                       "parallelism": f"tiles sharded 1 process per GPU x{world}, no data-path collective"},
            "gtexels_per_s": n * n * frames_per_s * 1e-9,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "speedup_vs_cpu_baseline": (frames_per_s / cpu["value"]) if cpu else None,
            "gather": gather,
            "extra": extra,
        }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
