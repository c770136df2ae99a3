"""Child process of tests/test_api_state_gpu.py: imports a dma-buf it inherited (memory another process owns) with
ocean_bind_output_dmabuf and synthesises one frame straight into it.
usage: dmabuf_writer.py <fd> <bytes> <disp_offset> <nrm_offset> <N> <tiles> <seed> <t>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import watersurfacerendering_amd as W

fd, nbytes, doff, noff, n, tiles, seed = (int(x) for x in sys.argv[1:8])
t = float(sys.argv[8])
b = W.OceanBatch(n, tiles, 0)
b.prepare(seed)
b.bind_output_dmabuf(fd, nbytes, doff, noff)
amp = b.compute_waves(t)
b.synchronize()
print("WRITER_OK", " ".join(repr(float(a)) for a in amp), flush=True)
b.bind_output(None, None)
b.close()
