#!/usr/bin/env python3
"""Developer probe for the slow window of the serial z pass: does it follow the shader clock?  A thread samples the GPU's sclk / power (hwmon
freq1_input / power1_average, or pp_dpm_sclk's starred level) every 10 ms while the main thread alternates HEAVY pipelined load (to provoke
power management) with windows of 50 serial 2048^2 frames whose per-kernel times are recorded; prints each window beside the clock and power
seen during it.    python3 tools/clock_watch.py [rounds]"""
import glob
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: E402,F401
import watersurfacerendering_amd as W  # noqa: E402


def find(pattern):
    hits = sorted(glob.glob(pattern))
    return hits[0] if hits else None


freq_file = find("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")
power_file = find("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") or find("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")
dpm_file = find("/sys/class/drm/card*/device/pp_dpm_sclk")
print("sources:", freq_file, power_file, dpm_file)
samples, stop = [], False


def read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def sampler():
    while not stop:
        f = read(freq_file) if freq_file else None
        p = read(power_file) if power_file else None
        d = None
        if dpm_file:
            txt = read(dpm_file) or ""
            star = [l for l in txt.splitlines() if l.endswith("*")]
            d = star[0] if star else None
        samples.append((time.perf_counter(), f, p, d))
        time.sleep(0.01)


th = threading.Thread(target=sampler, daemon=True)
th.start()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
heavy = W.OceanBatch(2048, 1, 0)
heavy.set_pipeline_depth(3)
heavy.prepare(1)
b = W.OceanBatch(2048, 1, 0)
b.prepare(2)
for r in range(rounds):
    if r % 2 == 1:
        t0 = time.perf_counter()
        heavy.time_frames(0.0, 0.05, 0, 12000, per_kernel=False)          # ~0.55 s of pipelined frames
        print(f"   (heavy pipelined load for {time.perf_counter() - t0:.2f} s)")
    for w in range(6):
        t0 = time.perf_counter()
        ms, k = b.time_frames(0.0, 0.05, 0, 50, per_kernel=True)
        t1 = time.perf_counter()
        seen = [s for s in samples if t0 <= s[0] <= t1] or samples[-1:]
        fr = [int(s[1]) / 1e6 for s in seen if s[1] and s[1].isdigit()]
        pw = [int(s[2]) / 1e6 for s in seen if s[2] and s[2].isdigit()]
        print(f"round {r} window {w}: z {k[0] * 1e3:6.2f} xb {k[1] * 1e3:6.2f} xd {k[2] * 1e3:6.2f} us   sclk {min(fr) if fr else None}-{max(fr) if fr else None} MHz   "
              f"power {max(pw) if pw else None} W   dpm {seen[-1][3]}", flush=True)
stop = True
heavy.close(); b.close()
