"""Runs tools/xb_trace.py in fresh processes, trace builds interleaved, until enough slow processes were seen; prints their output.
usage: xb_trace_hunt.py <max processes> <lib,lib> <wanted slow per lib>"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n, libs, want = int(sys.argv[1]), sys.argv[2].split(","), int(sys.argv[3])
got = {L: 0 for L in libs}; tried = {L: 0 for L in libs}
for p in range(n):
    for L in libs:
        if got[L] >= want: continue
        env = dict(os.environ, OCEAN_HIP_LIB=os.path.join(root, "watersurfacerendering_amd", f"libocean_hip_{L}.so"))
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "xb_trace.py"), "none", "2"], capture_output=True, text=True, env=env, cwd=root).stdout
        tried[L] += 1
        if "SLOW" in out:
            got[L] += 1
            print(f"==== [{L}] process {p}")
            print("\n".join(l[:700] for l in out.splitlines() if "amdgpu.ids" not in l), flush=True)
    if all(got[L] >= want for L in libs): break
print("tried", tried, "slow", got)
