"""Fresh processes, libraries interleaved: how many run serial k_xpass_b slow (> 22.4 us at 2048^2)?  (profiles/r03_bimodal_probe.txt)
usage: xb_hunt.py <processes> <lib,lib,...>     lib = default or the NAME of watersurfacerendering_amd/libocean_hip_NAME.so"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n, libs = int(sys.argv[1]), sys.argv[2].split(",")
slow = {L: [] for L in libs}; times = {L: [] for L in libs}
for p in range(n):
    for L in libs:
        env = dict(os.environ)
        env.pop("OCEAN_HIP_LIB", None)
        if L != "default": env["OCEAN_HIP_LIB"] = os.path.join(root, "watersurfacerendering_amd", f"libocean_hip_{L}.so")
        cmd = [sys.executable, os.path.join(root, "tools", "kernel_times.py"), "2048", "1", "200"]
        out = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=root).stdout
        m = re.search(r"k_xpass_b\s+([0-9.]+)", out)
        if not m:
            print(f"[{L}] process {p}: no timing in output: {out[-200:]!r}"); continue
        xb = float(m.group(1)); times[L].append(xb)
        if xb > 22.4:
            slow[L].append(xb); print(f"[{L}] process {p} SLOW: " + out.strip().splitlines()[-1][:160], flush=True)
for L in libs:
    t = sorted(times[L])
    print(f"{L}: {len(slow[L])} slow of {len(t)}; k_xpass_b min / median / max {t[0]:.2f} / {t[len(t)//2]:.2f} / {t[-1]:.2f}")
