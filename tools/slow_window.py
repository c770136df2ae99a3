#!/usr/bin/env python3
"""Round 6: what is the slow window of the serial 2048^2 z pass?

The container hides the part's telemetry (sysfs: a constant 95 MHz / 240 W; a child process samples it at 50 Hz all the same, and
`rocm-smi` / `amd-smi` once, for the record), so the clock is measured where it cannot be hidden -- INSIDE the kernel: a diagnostic build
(`make -C watersurfacerendering_amd/csrc variant NAME=probe DEFS="-DOCEAN_CLOCKPROBE -DOCEAN_DEVELOPER"`) has thread 0 of every workgroup
of k_zpass_c1 read s_memtime (shader cycles) and s_memrealtime (100 MHz) at its start and behind its last store; per LAUNCH that gives
the duration (max end - min start) and the in-kernel clock (median over the workgroups of cycles / ticks x 100 MHz;
MI355X_MICROARCH.md "DVFS give-back" (6)) -- one sample per launch, i.e. every ~60 us, far beyond the 20 Hz asked for.

Sequence: fresh process -> serial frames (windows of 4000 launches) -> SOAK seconds of pipelined load (tools/soak.py's shape: 2048^2 depth 3,
plus 8 x 1024^2 depth 2) -> serial frames again, at once and after 1 / 5 s of idling.  Printed per window: duration and clock percentiles,
their correlation over the launches, the clock of the slow launches (> 1.07 x the window's median duration) against the others, per-XCD
clocks, and a 100-launch-bucket time series.

    OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_probe.so python3 tools/slow_window.py [soak_seconds] [windows_after]
"""
import ctypes as C
import glob
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: E402,F401
import watersurfacerendering_amd as W  # noqa: E402

LAUNCHES, WGS, NWG = 4096, 1032, 1025
SOAK_S = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
AFTER = int(sys.argv[2]) if len(sys.argv) > 2 else 4
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(OUT, exist_ok=True)

# ---- the child sampler: whatever the box lets an ordinary user read, at 50 Hz ------------------------------------------------
SAMPLER = r'''
import glob, sys, time
pats = ["/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input", "/sys/class/drm/card*/device/hwmon/hwmon*/freq2_input",
        "/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input",
        "/sys/class/drm/card*/device/hwmon/hwmon*/temp1_input", "/sys/class/drm/card*/device/hwmon/hwmon*/temp2_input",
        "/sys/class/drm/card*/device/hwmon/hwmon*/temp3_input", "/sys/class/drm/card*/device/pp_dpm_sclk",
        "/sys/class/drm/card*/device/pp_dpm_mclk", "/sys/class/drm/card*/device/pp_dpm_fclk", "/sys/class/drm/card*/device/gpu_busy_percent"]
files = [f for p in pats for f in sorted(glob.glob(p))[:1]]
out = open(sys.argv[1], "w")
out.write("# t " + " ".join(files) + "\n")
while True:
    row = []
    for f in files:
        try:
            txt = open(f).read().strip()
            if "\n" in txt:
                star = [l for l in txt.splitlines() if l.endswith("*")]
                txt = star[0].replace(" ", "") if star else "?"
            row.append(txt)
        except OSError:
            row.append("-")
    out.write(f"{time.time():.3f} " + " ".join(row) + "\n"); out.flush()
    time.sleep(0.02)
'''
sysfs_log = os.path.join(OUT, "slow_window_sysfs.txt")
child = subprocess.Popen([sys.executable, "-c", SAMPLER, sysfs_log])
for tool in (["rocm-smi", "--showclocks", "--showpower", "--showtemp"], ["amd-smi", "metric", "--clock", "--power", "--temperature"]):
    try:
        r = subprocess.run(tool, capture_output=True, text=True, timeout=30)
        print(f"$ {' '.join(tool)}  (exit {r.returncode})\n" + "\n".join((r.stdout + r.stderr).splitlines()[:40]), flush=True)
    except Exception as exc:  # noqa: BLE001
        print(f"$ {' '.join(tool)}: {type(exc).__name__}: {exc}", flush=True)

L = W._abi.lib()
try:
    probe = L.ocean_debug_clockprobe
except AttributeError:
    raise SystemExit("needs the probe build: OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_probe.so")
probe.restype = C.c_int
probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t]

b = W.OceanBatch(2048, 1, 0)
b.prepare(0x5EED0000)
W._abi.check(probe(b._h, 1, None, 0, 0), "ocean_debug_clockprobe")
T0 = time.time()
frame = [0]          # frames this context has run = the chain's sequence number of the last one


def addresses(ctx, label):
    """developer builds: device addresses of the context's buffers (ocean_debug_buffers) -- alignment against 2 MiB / 1 GiB"""
    try:
        fn = L.ocean_debug_buffers
    except AttributeError:
        return
    fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    out = (C.c_void_p * 8)()
    if fn(ctx._h, 0, out) == 0:
        names = ("h0", "omega_q", "z", "zh", "hraw", "minmax", "disp", "nrm")
        print(f"   buffers of {label}: " + "  ".join(f"{n} {int(v or 0):#x} (mod 2 MiB {int(v or 0) % (2 << 20):#x})" for n, v in zip(names, out)))


def window(label, nframes=4000, ctx=None, counter=None):
    """nframes serial frames (depth 1, one stream: each z pass has the device to itself between its x passes), then the probe records."""
    ctx = b if ctx is None else ctx
    counter = frame if counter is None else counter
    t0 = time.time()
    first = counter[0] + 1
    for j in range(nframes):
        ctx.compute_waves_async(0.05 * j)
    ctx.synchronize()
    t1 = time.time()
    counter[0] += nframes
    buf = np.zeros((LAUNCHES, WGS, 4), dtype=np.uint64)
    W._abi.check(probe(ctx._h, 0, buf.ctypes.data_as(C.c_void_p), 0, buf.size), "ocean_debug_clockprobe")
    seqs = np.arange(first, first + nframes)
    rec = buf[seqs % LAUNCHES][:, :NWG, :]
    start, end, cyc, ids = (rec[..., k] for k in range(4))
    ok = (end > start).all(axis=1)
    dur = (end.max(axis=1) - start.min(axis=1)).astype(np.float64) / 100.0             # us (100 MHz ticks)
    ticks = (end - start).astype(np.float64)
    clk = np.median(cyc.astype(np.float64) / np.maximum(ticks, 1.0), axis=1) * 0.1     # GHz
    begin = (start.min(axis=1) - start.min()).astype(np.float64) / 100.0e6             # s since the window's first launch
    dur, clk, begin = dur[ok], clk[ok], begin[ok]
    med = float(np.median(dur))
    slow = dur > 1.07 * med
    xcc = (ids & np.uint64(0xF)).astype(np.int64)
    per_xcc = [float(np.median((cyc.astype(np.float64) / np.maximum(ticks, 1.0))[xcc == x])) * 0.1 if (xcc == x).any() else float("nan") for x in range(8)]
    p = lambda a, q: float(np.percentile(a, q))  # noqa: E731
    corr = float(np.corrcoef(dur, clk)[0, 1]) if dur.std() > 0 and clk.std() > 0 else float("nan")
    print(f"\n== {label}: {int(ok.sum())} launches in {t1 - t0:.2f} s (t = {t0 - T0:.1f} s)")
    print(f"   z-pass duration us   p05 {p(dur, 5):.2f}  median {med:.2f}  mean {dur.mean():.2f}  p95 {p(dur, 95):.2f}  max {dur.max():.2f}")
    print(f"   in-kernel clock GHz  p05 {p(clk, 5):.3f}  median {p(clk, 50):.3f}  p95 {p(clk, 95):.3f}  min {clk.min():.3f}")
    print(f"   slow launches (> 1.07 x median): {int(slow.sum())} ({100.0 * slow.mean():.1f} %)   clock of the slow ones {clk[slow].mean() if slow.any() else float('nan'):.3f} GHz, "
          f"of the others {clk[~slow].mean():.3f} GHz   corr(duration, clock) = {corr:+.3f}")
    print(f"   duration x clock (kilocycles per launch)  median {p(dur * clk, 50):.2f}  p05 {p(dur * clk, 5):.2f}  p95 {p(dur * clk, 95):.2f}   "
          f"slow ones {np.mean((dur * clk)[slow]) if slow.any() else float('nan'):.2f}  others {np.mean((dur * clk)[~slow]):.2f}")
    print("   per-XCC median clock GHz: " + " ".join(f"{v:.3f}" for v in per_xcc))
    # where the workgroups ran: per launch, workgroups per XCC and per compute unit (XCC id + SE / SH / CU bits 8..15 of HW_ID), and how long
    # each XCC was busy (its last end - the launch's first start)
    cu = (xcc << 8) | ((ids >> np.uint64(40)) & np.uint64(0xFF)).astype(np.int64)
    sample = np.arange(0, rec.shape[0], max(1, rec.shape[0] // 200))
    per_x = np.array([[int((xcc[l] == x).sum()) for x in range(8)] for l in sample])
    cu_counts = [np.unique(cu[l], return_counts=True)[1] for l in sample]
    t0l = start.min(axis=1)
    busy = np.array([[(float(end[l][xcc[l] == x].max()) - float(t0l[l])) / 100.0 if (xcc[l] == x).any() else 0.0 for x in range(8)] for l in sample])
    print(f"   workgroups per XCC (over {len(sample)} launches): min {per_x.min(axis=0).tolist()} max {per_x.max(axis=0).tolist()}   "
          f"compute units used {int(np.median([len(c) for c in cu_counts]))}, workgroups per unit max {int(max(c.max() for c in cu_counts))} "
          f"(median of the launches' maxima {int(np.median([c.max() for c in cu_counts]))})")
    print("   per-XCC busy time us (median over launches): " + " ".join(f"{v:.1f}" for v in np.median(busy, axis=0)))
    nb = max(1, len(dur) // 100)
    series = [(float(begin[i * 100]), float(np.median(dur[i * 100:(i + 1) * 100])), float(dur[i * 100:(i + 1) * 100].max()),
               float(np.median(clk[i * 100:(i + 1) * 100])), float(clk[i * 100:(i + 1) * 100].min())) for i in range(nb)]
    print("   buckets of 100 launches  [t s | median us | max us | median GHz | min GHz]:")
    for k in range(0, nb, 4):
        print("     " + "   ".join(f"{s[0]:6.3f} {s[1]:5.2f} {s[2]:5.2f} {s[3]:.3f} {s[4]:.3f}" for s in series[k:k + 4]))
    np.savez_compressed(os.path.join(OUT, f"slow_window_{label.split()[0]}.npz"), dur=dur, clk=clk, begin=begin)
    return med, float(np.median(clk))


addresses(b, "ctx0")
window("fresh0 (first serial frames of the process)")
addresses(b, "ctx0")
window("fresh1")
window("fresh2")
if SOAK_S <= 0:          # quick mode (A/B of kernel variants by cycles per launch): three windows, no soak
    if os.environ.get("OCEAN_REALLOC_TEST"):        # is the slow state a property of a context's ALLOCATION?  More contexts, the earlier ones kept alive
        keep = []
        for k in range(int(os.environ["OCEAN_REALLOC_TEST"])):
            if k % 2 == 1:                          # (every other one behind a 96 MiB allocation that shifts what the next context gets)
                import torch
                keep.append(torch.empty(96 << 20, dtype=torch.uint8, device="cuda"))
            c2 = W.OceanBatch(2048, 1, 0); c2.prepare(0x5EED0000)
            W._abi.check(probe(c2._h, 1, None, 0, 0), "ocean_debug_clockprobe")
            c2.compute_waves(0.0)
            cnt = [1]
            addresses(c2, f"ctx{k + 1}")
            window(f"ctx{k + 1} (another context, the earlier ones alive) warm-up", 1000, c2, cnt)
            window(f"ctx{k + 1} (another context, the earlier ones alive)", 3000, c2, cnt)
            keep.append(c2)
        window("ctx0 again (the first context)", 3000)
    if os.environ.get("OCEAN_XCD_ROT_SWEEP"):       # developer build: which XCD writes which column group, rotated window by window
        for rot in (1, 2, 3, 4, 5, 6, 0, 3, 0):
            os.environ["OCEAN_XCD_ROT"] = str(rot)
            window(f"rot{rot} (column groups of XCDs 1..7 rotated by {rot})", 2000)
    b.close(); child.terminate(); sys.exit(0)
# ---- sustained load -------------------------------------------------------------------------------------------------------
heavy = W.OceanBatch(2048, 1, 0); heavy.prepare(1); heavy.set_pipeline_depth(3)
heavy2 = W.OceanBatch(1024, 8, 0); heavy2.prepare(2); heavy2.set_pipeline_depth(2)
t0 = time.time(); n = 0
while time.time() - t0 < SOAK_S:
    for j in range(2000):
        heavy.compute_waves_async(0.01 * j)
        if j % 4 == 0:
            heavy2.compute_waves_async(0.01 * j)
    heavy.synchronize(); heavy2.synchronize(); n += 2000
print(f"\n-- soak: {n} pipelined 2048^2 frames + {n // 4} batches of 8 x 1024^2 in {time.time() - t0:.1f} s", flush=True)
for k in range(AFTER):
    window(f"after{k} (right behind the soak)" if k == 0 else f"after{k}")
time.sleep(1.0)
window("idle1 (after 1 s of idling)")
time.sleep(5.0)
window("idle5 (after 5 more s of idling)")
heavy.close(); heavy2.close(); b.close()
child.terminate()
rows = [l for l in open(sysfs_log) if not l.startswith("#")]
print(f"\n-- sysfs sampler: {len(rows)} samples at 50 Hz; header and distinct rows (time column dropped):")
print(open(sysfs_log).readline().strip())
seen = {}
for l in rows:
    seen.setdefault(" ".join(l.split()[1:]), 0)
    seen[" ".join(l.split()[1:])] += 1
for k, v in list(seen.items())[:12]:
    print(f"   {v:6d} x  {k}")
