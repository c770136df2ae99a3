"""Timeline of a burst of K pipelined frames between two drains (developer tool; what bench.py --steps 20 times per region).
  run:      rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bt -- python3 tools/burst_trace.py run [K] [depth]
  analyse:  python3 tools/burst_trace.py show <kernel_trace.csv> [K] [first frame] [frames]
Prints, for the median burst: first start -> last end, and per frame the start offset and duration of its three kernels."""
import csv, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def run(K, depth):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import devlib  # noqa: F401
    import watersurfacerendering_amd as W
    b = W.OceanBatch(2048, 1, 0); b.set_pipeline_depth(depth); b.prepare(1)
    for j in range(2000): b.compute_waves_async(0.016 * j)
    b.synchronize(); torch.cuda.synchronize()
    for rep in range(12):
        for j in range(K): b.compute_waves_async(0.016 * j)
        b.synchronize(); torch.cuda.synchronize()
    b.close()

def show(path, K, first=0, count=None):
    rows = []
    for r in csv.DictReader(open(path)):
        m = re.search(r"(k_[a-z_0-9]+)<", r["Kernel_Name"])
        if not m or not (m.group(1).startswith("k_zpass") or m.group(1).startswith("k_xpass")):
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "z" if "zpass" in m.group(1) else ("b" if m.group(1) == "k_xpass_b" else "d")))
    rows.sort()
    rows = rows[-12 * 3 * K:]
    bursts = [rows[i * 3 * K:(i + 1) * 3 * K] for i in range(12)]
    spans = sorted((max(e for _, e, _ in bu) - bu[0][0], i) for i, bu in enumerate(bursts))
    print("burst spans us:", [round(s * 1e-3, 1) for s, _ in spans], "-> per frame", round(spans[6][0] * 1e-3 / K, 2))
    bu = bursts[spans[6][1]]; t0 = bu[0][0]
    seq = {"z": [], "b": [], "d": []}
    for s, e, k in bu: seq[k].append(((s - t0) * 1e-3, (e - s) * 1e-3))
    prev_end = 0.0
    for f in range(first, K if count is None else min(K, first + count)):
        z, x, d = seq["z"][f], seq["b"][f], seq["d"][f]
        end = d[0] + d[1]
        print(f"frame {f:2d}: z @{z[0]:7.1f} {z[1]:5.1f} | b @{x[0]:7.1f} {x[1]:5.1f} | d @{d[0]:7.1f} {d[1]:5.1f} -> end {end:7.1f} (+{end - prev_end:5.1f})")
        prev_end = end

if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 20, int(sys.argv[3]) if len(sys.argv) > 3 else 3)
    else:
        show(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 20, int(sys.argv[4]) if len(sys.argv) > 4 else 0, int(sys.argv[5]) if len(sys.argv) > 5 else None)
