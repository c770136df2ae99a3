#!/usr/bin/env python3
"""Round 6: WHICH buffer's placement makes a context slow.  ocean_prepare's placement search with only some of the buffers differing between
candidates (developer switch OCEAN_PLACEMENT_MASK: bit 0 spectrum, 1-2 dispersion, 3-5 chain 0's intermediates, 6 its maps) and every
candidate's serial frame time printed (OCEAN_PLACEMENT_TRACE; both switches exist in developer builds only: make -C watersurfacerendering_amd/csrc variant
NAME=dev DEFS=-DOCEAN_DEVELOPER, then OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_dev.so): the mask whose candidates spread like the all-buffers search is the culprit.
    python3 tools/placement_attribution.py [N] [tiles] [candidates] [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: E402,F401
import watersurfacerendering_amd as W  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cands = int(sys.argv[3]) if len(sys.argv) > 3 else 12
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
os.environ["OCEAN_PLACEMENT_TRACE"] = "1"
MASKS = [(0x01, "spectrum"), (0x06, "dispersion"), (0x38, "intermediates"), (0x40, "maps"), (0x7f, "all seven"), (0x1f, "shipped group")]
print(f"{n}^2 x {tiles} tile(s), {cands} candidates per search, serial frame us: min / median / max  (spread)", flush=True)
for r in range(reps):
    for mask, name in MASKS:
        os.environ["OCEAN_PLACEMENT_MASK"] = hex(mask)
        b = W.OceanBatch(n, tiles, 0)
        b.set_placement_search(cands)
        sys.stderr.flush()
        b.prepare(0x5EED0000 + r)
        tried, chosen, worst = b.placement_report()
        print(f"  repeat {r}  {name:14s} (mask {mask:#04x}): candidates {tried:2d}  fastest {chosen:7.2f}  slowest {worst:7.2f}  spread {worst - chosen:6.2f} us", flush=True)
        b.close()
