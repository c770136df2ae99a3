"""Builds the two machine-readable summaries bench.py reads from committed rocprofv3 output (developer tool):

  profiles/kernel_stats.json   average duration per kernel from `rocprofv3 --kernel-trace --stats` CSVs of SERIAL frames
  profiles/traffic.json        HBM bytes per launch from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes

usage: profile_summaries.py stats <kernel_stats.csv> [tiles] ...   (kernels keyed name@N, or name@NxT for T tiles)
       profile_summaries.py traffic <pmc_summary.txt> <N> [tiles] [fetch_factor] [depth]
       (depth > 1: counter passes of PIPELINED frames, tools/pmc_depth.sh -- keyed name@N:depthD, what bench.py's roofline.pipelined_traffic quotes)
Entries are merged into the existing JSON files.  FETCH_SIZE is multiplied by `fetch_factor` (default 2.0: gfx950
tallies 128-byte read requests at 64 bytes, MI355X_MICROARCH.md HBM section; tools/ubench/fetchcal.hip calibrates the
factor for the access widths of these kernels, see profiles/README.md); WRITE_SIZE is taken as read.
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ("k_zpass", "k_xpass_b", "k_xpass_disp", "k_xfused", "k_phase_table")


def load(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return {}


def save(name, d):
    # the hash of the kernel sources these figures were measured with: bench.py quotes them only while it matches
    sys.path.insert(0, ROOT)
    import bench
    sha = bench.kernel_source_sha16()
    if d.get("_kernel_source_sha16") not in (None, sha):
        d = {k: v for k, v in d.items() if k.startswith("_") or v.get("kernel_source_sha16") == sha}     # drop entries of older kernels
    d["_kernel_source_sha16"] = sha
    json.dump(d, open(os.path.join(ROOT, "profiles", name), "w"), indent=1, sort_keys=True)


def rel(p):
    return os.path.relpath(os.path.abspath(p), ROOT)


def _sha():
    sys.path.insert(0, ROOT)
    import bench
    return bench.kernel_source_sha16()


def kernel_of(full):
    """(launch name, tile size) of a rocprofv3 kernel name.  The z pass has several kernel forms (k_zpass, k_zpass_c1: ocean_kernels.h);
    all of them are the frame's first launch, "k_zpass" in ocean_kernel_name's and bench.py's accounting."""
    m = re.search(r"(k_[a-z_0-9]+)<(\d+)", full)
    if not m:
        return (None, None)
    name = m.group(1)
    return ("k_zpass" if name.startswith("k_zpass") else name, int(m.group(2)))


def stats(path, tiles=1):
    out = load("kernel_stats.json")
    acc = {}
    for row in csv.DictReader(open(path)):
        k, n = kernel_of(row["Name"])
        if k not in NAMES:
            continue
        a = acc.setdefault((k, n), [0, 0.0])
        a[0] += int(row["Calls"]); a[1] += float(row["TotalDurationNs"])
    for (k, n), (calls, total) in acc.items():
        key = f"{k}@{n}" + (f"x{tiles}" if tiles > 1 else "")
        out[key] = {"avg_us": total / calls * 1e-3, "calls": calls, "source": rel(path) + " (rocprofv3 --kernel-trace --stats, serial frames)",
                    "kernel_source_sha16": _sha()}
    save("kernel_stats.json", out)


def traffic(path, n, tiles=1, fetch_factor=2.0, depth=1):
    out = load("traffic.json")
    cur = None
    vals = {}
    for line in open(path):
        if not line.startswith(" "):
            cur = kernel_of(line.strip())[0]
            continue
        m = re.match(r"\s+(\S+)\s+mean\s+([0-9.]+)", line)
        if m and cur in NAMES:
            vals.setdefault(cur, {})[m.group(1)] = float(m.group(2))
    for k, v in vals.items():
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        key = f"{k}@{n}" + (f"x{tiles}" if tiles > 1 else "") + (f":depth{depth}" if depth > 1 else "")
        ent = {"hbm_bytes_per_launch": int(v["FETCH_SIZE"] * 1024 * fetch_factor + v["WRITE_SIZE"] * 1024),
               "fetch_size_kb_raw": v["FETCH_SIZE"], "write_size_kb_raw": v["WRITE_SIZE"], "fetch_factor": fetch_factor,
               "source": rel(path) + (" (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, tools/pmc3.sh)" if depth == 1 else
                                      f" (the same counters over frames at pipeline depth {depth}: non-temporal map stores, {depth} chains rotating "
                                      "through the memory-side cache; tools/pmc_depth.sh)"),
               "kernel_source_sha16": _sha()}
        if k == "k_zpass":
            ent["correction"] = (f"FETCH_SIZE x {fetch_factor}: its loads are coalesced (16 / 4 bytes per lane), for which "
                                 "tools/ubench/fetchcal.hip measures known bytes / FETCH_SIZE = 2.000 (128-byte requests tallied at 64: "
                                 "MI355X_MICROARCH.md HBM section; profiles/r02_fetch_calibration.txt); WRITE_SIZE as read")
        else:       # the x passes read 32-byte halves of 64-byte pieces out of dense runs (row-blocked intermediates)
            ent["hbm_bytes_per_launch_low"] = int(v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024)
            ent["correction"] = (f"FETCH_SIZE x {fetch_factor} as for coalesced reads.  With the row-blocked intermediates these kernels read dense "
                                 "runs (a workgroup uses 32 bytes of every 64-byte piece, its neighbour on the same XCD the other 32), and the "
                                 "x 2 reading lands within 10 % of the algorithmic bytes; with the column-major layout of the first half of the "
                                 "round (32-byte pieces at a 16.5 KB stride) the raw counter alone was already 1.4-1.9 x the useful bytes "
                                 "(profiles/r02_fetch_calibration.txt).  hbm_bytes_per_launch_low = the x 1 reading, kept as the lower bound; "
                                 "WRITE_SIZE as read")
        out[key] = ent
    save("traffic.json", out)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    elif sys.argv[1] == "traffic":
        traffic(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 1, float(sys.argv[5]) if len(sys.argv) > 5 else 2.0,
                int(sys.argv[6]) if len(sys.argv) > 6 else 1)
    else:
        raise SystemExit(__doc__)
