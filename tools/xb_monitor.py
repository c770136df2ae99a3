"""Long-running monitor for the transient slow state of serial k_xpass_b (profiles/r03_bimodal_probe.txt section 4): one context, bursts of 100 serial
2048^2 frames with per-kernel times back to back for T seconds, each burst stamped with what the driver reports for THIS device (found by its PCI bus
id): package power, temperatures, sclk / mclk (hwmon), DPM levels, memory-busy percentage.  Prints a line per second, every burst whose k_xpass_b
is slow (> 22.4 us), and at the end the mean readings of slow against normal bursts.
usage: xb_monitor.py [seconds]"""
import ctypes as C, glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
T0 = time.perf_counter()
import watersurfacerendering_amd as W
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
hip = C.CDLL("libamdhip64.so")
buf = C.create_string_buffer(64); hip.hipDeviceGetPCIBusId(buf, 64, 0)
bus = buf.value.decode().lower()
card = next((c for c in sorted(glob.glob("/sys/class/drm/card*/device")) if bus in os.path.realpath(c).lower()), None)
hw = (glob.glob(card + "/hwmon/hwmon*") or [None])[0] if card else None

def rd(path):
    try:
        return open(path).read().strip()
    except OSError:
        return ""
def cur(path):
    return next((l.split(":")[1].strip().rstrip("*").strip() for l in rd(path).splitlines() if "*" in l), "?")
def sample():
    s = {}
    if hw:
        s["W"] = int(rd(hw + "/power1_input") or 0) / 1e6
        for k in ("temp1_input", "temp2_input", "temp3_input"):
            v = rd(hw + "/" + k)
            if v: s[k[:5]] = int(v) / 1000
        for k in ("freq1_input", "freq2_input"):
            v = rd(hw + "/" + k)
            if v: s[k[:5]] = int(v) / 1e6
    if card:
        s["sclk"] = cur(card + "/pp_dpm_sclk"); s["mclk"] = cur(card + "/pp_dpm_mclk"); s["fclk"] = cur(card + "/pp_dpm_fclk")
        s["membusy"] = rd(card + "/mem_busy_percent"); s["busy"] = rd(card + "/gpu_busy_percent")
    return s
print("device", bus, "card", card, "first sample", sample(), flush=True)
b = W.OceanBatch(2048, 1, 0); b.prepare(0x5EED0000)
rows, last_print = [], 0.0
while time.perf_counter() - T0 < secs:
    ms, k = b.time_frames(0.0, 0.05, 0, 100)
    t = time.perf_counter() - T0
    s = sample()
    xb = k[1] * 1e3
    rows.append((t, k[0] * 1e3, xb, k[2] * 1e3, s))
    slow = xb > 22.4
    if slow or t - last_print >= 1.0:
        if not slow: last_print = t
        print(f"{t:7.2f} s {'SLOW' if slow else '    '} z {k[0]*1e3:6.2f} xb {xb:6.2f} disp {k[2]*1e3:6.2f}  " +
              " ".join(f"{a}={v:.1f}" if isinstance(v, float) else f"{a}={v}" for a, v in s.items()), flush=True)
b.close()
slow = [r for r in rows if r[2] > 22.4]; norm = [r for r in rows if r[2] <= 22.4]
print(f"bursts {len(rows)}, slow {len(slow)}" + (f" (first at {slow[0][0]:.2f} s, last at {slow[-1][0]:.2f} s)" if slow else ""))
for name, grp in (("slow", slow), ("normal", norm)):
    if grp:
        keys = [k for k, v in grp[0][4].items() if isinstance(v, float)]
        print(f"  {name:6s} xb {sum(r[2] for r in grp)/len(grp):6.2f}  z {sum(r[1] for r in grp)/len(grp):6.2f}  disp {sum(r[3] for r in grp)/len(grp):6.2f}  " +
              " ".join(f"{k}={sum(r[4][k] for r in grp)/len(grp):.1f}" for k in keys))
