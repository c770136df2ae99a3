#!/bin/bash
# usage: tools/bimodal_probe.sh <processes> [libs...]   -- per-kernel times of serial 2048^2 frames in <processes> FRESH processes per
# library, interleaved; prints every line and, per library, how many processes saw k_xpass_b more than 15 % above that library's median
# (the "slow mode" of round 2: 33 instead of 25.5 us in some processes).
P=${1:-10}; shift
LIBS=${@:-default}
for i in $(seq 1 $P); do
  for L in $LIBS; do
    if [ "$L" = "default" ]; then unset OCEAN_HIP_LIB; else export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_$L.so; fi
    echo "[$L] $(python tools/kernel_times.py 2048 1 200 2>/dev/null | grep N=)"
  done
done | tee /tmp/bimodal_probe.txt
python3 - <<'PY'
import re, statistics
rows = {}
for l in open('/tmp/bimodal_probe.txt'):
    m = re.match(r"\[(\S+)\].*k_xpass_b\s+([0-9.]+)", l)
    if m: rows.setdefault(m.group(1), []).append(float(m.group(2)))
for lib, v in rows.items():
    med = statistics.median(v)
    print(f"{lib}: {len(v)} processes, k_xpass_b median {med:.2f} us, min {min(v):.2f}, max {max(v):.2f}, slow (> 1.15 x median): {sum(x > 1.15 * med for x in v)}")
PY
