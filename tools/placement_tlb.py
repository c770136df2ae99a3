#!/usr/bin/env python3
"""Round 6: do the slow allocations miss more in the address-translation caches?  One 2048^2 context whose Prepare times 12 candidate placements
(OCEAN_PLACEMENT_TRACE prints their serial frame times: a developer build, -DOCEAN_DEVELOPER, through OCEAN_HIP_LIB); run under `rocprofv3 --pmc <translation counters>` the per-dispatch counters of the
z pass can be grouped by candidate afterwards (tools/placement_tlb.py --summarise <counter_collection.csv>: the last 12 x 48 z-pass dispatches).
    rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum --output-format csv -d out -- python3 tools/placement_tlb.py"""
import csv
import os
import sys

CANDS, FRAMES = 12, 48
if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "k_zpass" in r["Kernel_Name"]]
    names = sorted({r["Counter_Name"] for r in rows})
    for name in names:
        v = [float(r["Counter_Value"]) for r in sorted((r for r in rows if r["Counter_Name"] == name), key=lambda r: int(r["Dispatch_Id"]))]
        v = v[-CANDS * FRAMES:]
        per = [sum(v[k * FRAMES + 8:(k + 1) * FRAMES]) / (FRAMES - 8) for k in range(CANDS)]
        print(f"{name}: per launch, by candidate: " + " ".join(f"{x:.0f}" for x in per))
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: E402,F401
import watersurfacerendering_amd as W  # noqa: E402

os.environ["OCEAN_PLACEMENT_TRACE"] = "1"
for rep in range(2):
    b = W.OceanBatch(2048, 1, 0)
    b.set_placement_search(CANDS)
    b.prepare(0x5EED0000 + rep)
    print("report", b.placement_report(), flush=True)
    if rep == 0:
        keep = b          # (stays alive: the second context draws other memory)
keep.close(); b.close()
