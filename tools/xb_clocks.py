"""The slow serial k_xpass_b against time and the device's clock levels.  One context; per-kernel times of short bursts of serial 2048^2 frames,
each line stamped with the time since process start and the DPM levels the driver reports (sysfs pp_dpm_*; rocm-smi is not used: it would
be a second process on the device).  Phases: back-to-back bursts, bursts with idle gaps, back-to-back again.
usage: xb_clocks.py"""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
T0 = time.perf_counter()
import watersurfacerendering_amd as W

def levels():
    out = []
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        for f in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
            try:
                cur = [l.strip() for l in open(os.path.join(card, f)) if "*" in l]
                out.append(f"{f[7:]}={cur[0].replace(' ', '') if cur else '?'}")
            except OSError:
                pass
        try:
            out.append("busy=" + open(os.path.join(card, "gpu_busy_percent")).read().strip())
        except OSError:
            pass
        break
    return " ".join(out) if out else "(no sysfs clocks readable)"

print("at start:", levels(), flush=True)
b = W.OceanBatch(2048, 1, 0); b.prepare(0x5EED0000)
def burst(tag, frames=100):
    ms, k = b.time_frames(0.0, 0.05, 0, frames)
    print(f"{time.perf_counter() - T0:7.3f} s {tag:10s} " + " ".join(f"{v*1e3:6.2f}" for v in k) + f"  frame {ms/frames*1e3:6.1f}   {levels()}", flush=True)
for i in range(40): burst("busy")
for gap in (0.01, 0.05, 0.2, 1.0):
    for i in range(6):
        time.sleep(gap); burst(f"gap{gap}")
for i in range(10): burst("busy")
b.close()
