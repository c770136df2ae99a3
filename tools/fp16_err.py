import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import watersurfacerendering_amd as W
from oracle import oracle as O
for n in (512, 2048):
    b = W.OceanBatch(n, 1, 0); b.set_spectrum_precision(16); b.prepare(7)
    xi = b.read_xi(0); o = O.Oracle(n); o.prepare(xi=xi)
    for t in (0.0, 4.5, 1000.0):
        ao, do, no = o.compute_waves(t, fft=O.FFT_F64)
        a = float(b.compute_waves(t)[0]); d, q = b.read_maps()
        errs = [float(np.abs(d[0][..., c] - do[..., c]).max() / np.abs(do[..., c]).max()) for c in range(3)] + \
               [float(np.abs(q[0][..., c] - no[..., c]).max() / np.abs(no[..., c]).max()) for c in range(4)]
        print(n, t, "amp rel", abs(a - ao) / ao, "chan", ["%.1e" % e for e in errs])
    ms, k = b.time_frames(0.0, 0.05, 10, 100)
    print(n, "fp16 spectrum us/frame", ms / 100 * 1e3, [x * 1e3 for x in k])
    b.close()
