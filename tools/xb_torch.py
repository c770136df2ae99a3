"""Reproducer hunt for the slow-queue state of k_xpass_b: does creating a torch stream (a second user of the process's hardware queues) before /
after the context make the context's queue a slow one?  usage: xb_torch.py <mode>   mode: none | import | stream_before | stream_after | two_streams_before"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
if mode != "none":
    import torch
    if mode in ("stream_before", "two_streams_before"):
        s1 = torch.cuda.Stream()
        if mode == "two_streams_before": s2 = torch.cuda.Stream()
import watersurfacerendering_amd as W
b = W.OceanBatch(2048, 1, 0); b.prepare(1)
if mode == "stream_after":
    s1 = torch.cuda.Stream()
out = []
for rep in range(3):
    ms, k = b.time_frames(0.0, 0.05, 100, 300)
    out.append(f"{k[0]*1e3:.2f}/{k[1]*1e3:.2f}/{k[2]*1e3:.2f}")
print(f"{mode:20s} z/xb/disp: " + "  ".join(out), flush=True)
b.close()
