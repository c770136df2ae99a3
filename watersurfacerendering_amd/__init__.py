"""watersurfacerendering_amd -- MI355X-native Tessendorf FFT ocean synthesiser.

Host-side mirror of the reference's `class WSTessendorf`
(/root/reference/src/scene/WSTessendorf.h:58-122) over the C ABI of
include/ocean.h (libocean_hip.so: hand-written HIP for gfx950).  Same method
names, argument meaning and (absence of) error behaviour as the reference class
so parity tests read like calls on the original object:

    ws = WSTessendorf(512, 1000.0)
    ws.SetWindDirection((1.0, 0.0)); ws.SetWindSpeed(10.0)
    ws.Prepare(seed=7)
    A = ws.ComputeWaves(1.5)
    disp, normals = ws.GetDisplacements(), ws.GetNormals()     # (N, N, 4) float32

`OceanBatch` exposes the batched / asynchronous surface (T independent tiles
per context, device-resident maps, event timing) the bench and the multi-GPU
path use.  Nothing here falls back to a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import math

from typing import Optional

import numpy as np

from . import _abi
from ._abi import OceanError, Params, build  # noqa: F401

__all__ = ["WSTessendorf", "OceanBatch", "OceanError", "build", "host_register", "host_unregister", "comm_unique_id"]


def _is_pow2(n: int) -> bool:
    return n > 0 and (n & (n - 1)) == 0


class OceanBatch:
    """T independent N x N tiles on one device (ocean_create ... ocean_destroy)."""

    def __init__(self, tile_size: int = 512, tiles: int = 1, device: int = 0):
        self._L = _abi.lib()
        self._h = C.c_void_p()
        _abi.check(self._L.ocean_create(C.byref(self._h), tile_size, tiles, device), "ocean_create")
        self.tiles = tiles
        self.device = device

    # -- lifetime ---------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.ocean_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- properties ---------------------------------------------------------------
    @property
    def tile_size(self) -> int:
        return int(self._L.ocean_tile_size(self._h))

    def get_params(self, tile: int = 0) -> Params:
        p = Params()
        _abi.check(self._L.ocean_get_params(self._h, tile, C.byref(p)), "ocean_get_params")
        return p

    def set_params(self, tile: int = _abi.OCEAN_ALL_TILES, **kw):
        """Patch the given fields of one tile, or of EVERY tile (each keeps its other parameters)."""
        tiles = range(self.tiles) if tile == _abi.OCEAN_ALL_TILES else (tile,)
        for i in tiles:
            p = self.get_params(i)
            for k, v in kw.items():
                k = "lambda_" if k in ("lambda", "lam") else k
                if not hasattr(p, k):
                    raise TypeError(f"unknown ocean parameter {k!r}")
                setattr(p, k, v)
            _abi.check(self._L.ocean_set_params(self._h, i, C.byref(p)), "ocean_set_params")

    def set_lambda(self, lam: float, tile: int = _abi.OCEAN_ALL_TILES):
        _abi.check(self._L.ocean_set_lambda(self._h, tile, lam), "ocean_set_lambda")

    def set_tile_size(self, n: int):
        _abi.check(self._L.ocean_set_tile_size(self._h, n), "ocean_set_tile_size")

    # -- Prepare / ComputeWaves -----------------------------------------------------
    def prepare(self, seed: int = 0, xi: np.ndarray | None = None):
        ptr = None
        if xi is not None:
            n = self.tile_size
            xi = np.ascontiguousarray(xi, dtype=np.float32).reshape(self.tiles, n, n, 2)
            ptr = xi.ctypes.data_as(C.c_void_p)
        _abi.check(self._L.ocean_prepare(self._h, seed & 0xFFFFFFFFFFFFFFFF, ptr), "ocean_prepare")

    def compute_waves(self, t: float) -> np.ndarray:
        amp = np.empty(self.tiles, dtype=np.float32)
        _abi.check(self._L.ocean_compute_waves(self._h, t, amp.ctypes.data_as(C.POINTER(C.c_float))),
                   "ocean_compute_waves")
        return amp

    def compute_waves_read(self, t: float, disp: np.ndarray | None = None, nrm: np.ndarray | None = None):
        """ComputeWaves(t) and the read-out of every tile's maps as ONE blocking call (ocean_compute_waves_read: the normal map's copy
        runs beside the displacement pass).  Returns (amplitudes, disp, nrm); disp / nrm [tiles, N, N, 4] float32 are allocated unless
        passed in (pass page-locked arrays -- host_register -- for true DMAs)."""
        n = self.tile_size
        amp = np.empty(self.tiles, dtype=np.float32)
        d = np.empty((self.tiles, n, n, 4), dtype=np.float32) if disp is None else disp
        q = np.empty((self.tiles, n, n, 4), dtype=np.float32) if nrm is None else nrm
        assert d.dtype == np.float32 and q.dtype == np.float32 and d.size == q.size == self.tiles * n * n * 4 and d.flags.c_contiguous and q.flags.c_contiguous
        _abi.check(self._L.ocean_compute_waves_read(self._h, t, amp.ctypes.data_as(C.POINTER(C.c_float)), d.ctypes.data_as(C.c_void_p),
                                                    q.ctypes.data_as(C.c_void_p)), "ocean_compute_waves_read")
        return amp, d, q

    def set_placement_search(self, trials: int):
        """Candidates the next prepare() times for the spectrum + intermediates (ocean_set_placement_search): 0 = the library's rule, 1 = off."""
        _abi.check(self._L.ocean_set_placement_search(self._h, int(trials)), "ocean_set_placement_search")

    def placement_report(self):
        """(candidates timed by the most recent prepare(), serial frame us of the chosen one, of the slowest one)."""
        n, a, b = C.c_int(), C.c_float(), C.c_float()
        _abi.check(self._L.ocean_placement_report(self._h, C.byref(n), C.byref(a), C.byref(b)), "ocean_placement_report")
        return n.value, a.value, b.value

    @property
    def fault_recoveries(self) -> int:
        """How often the host re-ran frames because an in-launch wait had given up (ocean_fault_recoveries; 0 on a dedicated device)."""
        return int(self._L.ocean_fault_recoveries(self._h))

    def compute_waves_async(self, t: float):
        _abi.check(self._L.ocean_compute_waves_async(self._h, t), "ocean_compute_waves_async")

    def wait_frame(self) -> np.ndarray:
        """Amplitudes of the most recently enqueued frame once its last workgroup has finished (ocean_wait_frame: a poll of the
        frame's completion records, not a stream synchronisation; copies enqueued behind the frame may still be running)."""
        amp = np.empty(self.tiles, dtype=np.float32)
        _abi.check(self._L.ocean_wait_frame(self._h, amp.ctypes.data_as(C.POINTER(C.c_float))), "ocean_wait_frame")
        return amp

    def set_frame_tracking(self, on: bool):
        """Asynchronous frames leave completion records too, so that wait_frame() polls instead of synchronising the stream."""
        _abi.check(self._L.ocean_set_frame_tracking(self._h, int(bool(on))), "ocean_set_frame_tracking")

    def set_time_offsets(self, offsets):
        if offsets is None:
            _abi.check(self._L.ocean_set_time_offsets(self._h, None), "ocean_set_time_offsets")
            return
        o = np.ascontiguousarray(offsets, dtype=np.float32)
        assert o.size == self.tiles
        _abi.check(self._L.ocean_set_time_offsets(self._h, o.ctypes.data_as(C.c_void_p)), "ocean_set_time_offsets")

    def synchronize(self):
        _abi.check(self._L.ocean_synchronize(self._h), "ocean_synchronize")

    def heights(self, tile: int = 0):
        a, mn, mx = C.c_float(), C.c_float(), C.c_float()
        _abi.check(self._L.ocean_get_heights(self._h, tile, C.byref(a), C.byref(mn), C.byref(mx)),
                   "ocean_get_heights")
        return a.value, mn.value, mx.value

    # -- read-out -------------------------------------------------------------------
    def read_maps(self, first: int = 0, count: int | None = None):
        count = self.tiles - first if count is None else count
        n = self.tile_size
        d = np.empty((count, n, n, 4), dtype=np.float32)
        q = np.empty((count, n, n, 4), dtype=np.float32)
        _abi.check(self._L.ocean_read_maps(self._h, first, count, d.ctypes.data_as(C.c_void_p),
                                           q.ctypes.data_as(C.c_void_p)), "ocean_read_maps")
        return d, q

    def read_maps_async(self, disp: np.ndarray, nrm: np.ndarray, first: int = 0, count: int | None = None):
        """Enqueue the D2H copy of the last enqueued frame's maps into caller arrays (pin them with
        host_register for a true asynchronous DMA); valid after synchronize()."""
        count = self.tiles - first if count is None else count
        _abi.check(self._L.ocean_read_maps_async(self._h, first, count, disp.ctypes.data_as(C.c_void_p),
                                                 nrm.ctypes.data_as(C.c_void_p)), "ocean_read_maps_async")

    def read_maps_staging(self, staging: np.ndarray, vertices_bytes: int, indices_bytes: int, tile: int = 0) -> int:
        """The reference's staging upload (WaterSurfaceMesh.cpp:701-755): enqueue the copy of the last enqueued
        frame's maps of `tile` into a byte buffer laid out [vertices | indices | pad16 | displacements | normals].
        Returns the number of bytes the caller then flushes; valid after synchronize()."""
        assert staging.dtype == np.uint8 and staging.flags["C_CONTIGUOUS"]
        n = self.tile_size
        off = int(self._L.ocean_staging_map_offset(vertices_bytes, indices_bytes))
        if staging.nbytes < off + 2 * n * n * 16:
            raise ValueError("staging buffer too small")
        flush = C.c_size_t()
        _abi.check(self._L.ocean_read_maps_staging(self._h, tile, staging.ctypes.data_as(C.c_void_p), vertices_bytes,
                                                   indices_bytes, C.byref(flush)), "ocean_read_maps_staging")
        return int(flush.value)

    # -- multi-GPU gather of the packed maps (RCCL) ------------------------------------------
    def comm_init(self, nranks: int, rank: int, unique_id: bytes):
        assert len(unique_id) == _abi.OCEAN_COMM_ID_BYTES
        buf = C.create_string_buffer(unique_id, _abi.OCEAN_COMM_ID_BYTES)
        _abi.check(self._L.ocean_comm_init(self._h, nranks, rank, buf), "ocean_comm_init")

    def comm_count(self):
        """(ranks, rank) of the context's communicator as RCCL reports them."""
        n, r = C.c_int(), C.c_int()
        _abi.check(self._L.ocean_comm_count(self._h, C.byref(n), C.byref(r)), "ocean_comm_count")
        return n.value, r.value

    def comm_destroy(self):
        _abi.check(self._L.ocean_comm_destroy(self._h), "ocean_comm_destroy")

    def gather_maps(self, root: int, recv_disp: int | None, recv_nrm: int | None, half: bool = False):
        """Enqueue the RCCL gather of the last enqueued frame's maps to `root` (device pointers of the receive
        arrays [nranks][tiles][N][N][4] on the root, None elsewhere); asynchronous, see ocean.h.  half=True sends
        the maps as IEEE halves (receive arrays of float16)."""
        fn = self._L.ocean_gather_maps_f16 if half else self._L.ocean_gather_maps
        _abi.check(fn(self._h, root, C.c_void_p(recv_disp), C.c_void_p(recv_nrm)), "ocean_gather_maps")

    def device_maps(self):
        d, q = C.c_void_p(), C.c_void_p()
        _abi.check(self._L.ocean_device_maps(self._h, C.byref(d), C.byref(q)), "ocean_device_maps")
        return d.value, q.value

    def displace_grid(self, tile: int = 0, grid_size: Optional[int] = None, vertex_distance: Optional[float] = None,
                      uv_scale: float = 1.0, choppy: float = -1.0):
        """Vertex-stage consumer (WaterSurfaceMesh.vert:24-41 on the grid of WaterSurfaceMesh.cpp:500-533) of the
        most recent frame: returns (positions, normals), each ((grid_size+1)^2, 4) float32.  Defaults are the
        reference's: grid_size = tile size, vertex_distance = 1000/512 (WaterSurfaceMesh.h:199-202), choppy =
        the default lambda."""
        g = self.tile_size if grid_size is None else int(grid_size)
        vd = (1000.0 / 512.0) if vertex_distance is None else float(vertex_distance)
        _abi.check(self._L.ocean_displace_grid(self._h, tile, g, vd, uv_scale, choppy), "ocean_displace_grid")
        pos = np.empty(((g + 1) * (g + 1), 4), dtype=np.float32)
        nrm = np.empty_like(pos)
        _abi.check(self._L.ocean_read_grid(self._h, pos.ctypes.data_as(C.c_void_p), nrm.ctypes.data_as(C.c_void_p)),
                   "ocean_read_grid")
        return pos, nrm

    def displace_grid_cascades(self, uv_scales, first_tile: int = 0, grid_size: Optional[int] = None,
                               vertex_distance: Optional[float] = None, choppy: float = -1.0):
        """Sum of the tiles first_tile .. first_tile+len(uv_scales)-1 as cascades (ocean_displace_grid_cascades):
        returns (positions, normals) like displace_grid."""
        sc = np.ascontiguousarray(uv_scales, dtype=np.float32)
        g = self.tile_size if grid_size is None else int(grid_size)
        vd = (1000.0 / 512.0) if vertex_distance is None else float(vertex_distance)
        _abi.check(self._L.ocean_displace_grid_cascades(self._h, first_tile, sc.size, g, vd, sc.ctypes.data_as(C.POINTER(C.c_float)), choppy),
                   "ocean_displace_grid_cascades")
        pos = np.empty(((g + 1) * (g + 1), 4), dtype=np.float32)
        nrm = np.empty_like(pos)
        _abi.check(self._L.ocean_read_grid(self._h, pos.ctypes.data_as(C.c_void_p), nrm.ctypes.data_as(C.c_void_p)), "ocean_read_grid")
        return pos, nrm

    def build_mips(self, tile: int = 0):
        """Mip chain of both maps of `tile` (ocean_build_mips: the reference's s_kUseMipMapping path, Texture2D.cpp:228-330):
        returns (disp_levels, nrm_levels), lists of (N >> l, N >> l, 4) float32 arrays for l = 1 .. log2 N."""
        _abi.check(self._L.ocean_build_mips(self._h, tile), "ocean_build_mips")
        texels = int(self._L.ocean_mip_texels(self.tile_size))
        d = np.empty((texels, 4), dtype=np.float32)
        q = np.empty_like(d)
        _abi.check(self._L.ocean_read_mips(self._h, d.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p)), "ocean_read_mips")
        out_d, out_q, off, w = [], [], 0, self.tile_size // 2
        while w >= 1:
            out_d.append(d[off:off + w * w].reshape(w, w, 4)); out_q.append(q[off:off + w * w].reshape(w, w, 4))
            off += w * w; w //= 2
        return out_d, out_q

    def export_maps(self):
        """dma-buf of the current map set (ocean_export_maps): (fd, disp_offset, nrm_offset, bytes, map_set); close the fd yourself."""
        fd, ms = C.c_int(-1), C.c_int(0)
        do, no, nb = C.c_size_t(), C.c_size_t(), C.c_size_t()
        _abi.check(self._L.ocean_export_maps(self._h, C.byref(fd), C.byref(do), C.byref(no), C.byref(nb), C.byref(ms)), "ocean_export_maps")
        return fd.value, do.value, no.value, nb.value, ms.value

    def bind_output_dmabuf(self, fd: int, nbytes: int, disp_offset: int, nrm_offset: int):
        """Write the maps into memory another owner exported as a dma-buf (ocean_bind_output_dmabuf)."""
        _abi.check(self._L.ocean_bind_output_dmabuf(self._h, fd, nbytes, disp_offset, nrm_offset), "ocean_bind_output_dmabuf")

    def bind_output(self, d_disp: int | None, d_nrm: int | None):
        _abi.check(self._L.ocean_bind_output(self._h, C.c_void_p(d_disp), C.c_void_p(d_nrm)), "ocean_bind_output")

    def set_mode(self, mode: int):
        _abi.check(self._L.ocean_set_mode(self._h, mode), "ocean_set_mode")

    def set_dispersion(self, kind: int, param: float = 0.0):
        """0 deep water (reference), 1 finite depth (param = D), 2 capillary (param = L); next prepare()."""
        _abi.check(self._L.ocean_set_dispersion(self._h, kind, param), "ocean_set_dispersion")

    def set_spectrum_precision(self, bits: int):
        _abi.check(self._L.ocean_set_spectrum_precision(self._h, bits), "ocean_set_spectrum_precision")

    def set_intermediate_precision(self, bits: int):
        """32 (default) or 16: half2 intermediates between the two passes (config 4's reduced mode); next prepare()."""
        _abi.check(self._L.ocean_set_intermediate_precision(self._h, bits), "ocean_set_intermediate_precision")

    def set_pipeline_depth(self, depth: int):
        _abi.check(self._L.ocean_set_pipeline_depth(self._h, depth), "ocean_set_pipeline_depth")

    def set_merged_xpass(self, on: bool):
        _abi.check(self._L.ocean_set_merged_xpass(self._h, int(bool(on))), "ocean_set_merged_xpass")

    def set_start_ramp(self, on: bool):
        _abi.check(self._L.ocean_set_start_ramp(self._h, int(bool(on))), "ocean_set_start_ramp")

    def set_external_readers(self, on: bool):
        _abi.check(self._L.ocean_set_external_readers(self._h, int(bool(on))), "ocean_set_external_readers")

    @property
    def stream(self) -> int:
        return self._L.ocean_stream(self._h) or 0

    def set_stream(self, s: int | None):
        _abi.check(self._L.ocean_set_stream(self._h, C.c_void_p(s)), "ocean_set_stream")

    def select_streams(self, frames: int = 50):
        """Put the context's work on the fastest of the process's hardware queues (ocean_select_streams): times `frames` serial frames on
        each of its first four streams and re-orders them; returns the four frame times in microseconds, fastest first."""
        us = (C.c_float * 4)()
        _abi.check(self._L.ocean_select_streams(self._h, frames, us), "ocean_select_streams")
        return [float(x) for x in us]

    def read_spectrum(self, tile: int = 0):
        n = self.tile_size
        h0 = np.empty((n, n, 2), dtype=np.float32)
        om = np.empty((n, n), dtype=np.float32)
        _abi.check(self._L.ocean_read_spectrum(self._h, tile, h0.ctypes.data_as(C.c_void_p),
                                               om.ctypes.data_as(C.c_void_p)), "ocean_read_spectrum")
        return h0, om

    def read_xi(self, tile: int = 0):
        n = self.tile_size
        xi = np.empty((n, n, 2), dtype=np.float32)
        _abi.check(self._L.ocean_read_xi(self._h, tile, xi.ctypes.data_as(C.c_void_p)), "ocean_read_xi")
        return xi

    def time_frames(self, t0: float, dt: float, warmup: int, frames: int, per_kernel: bool = True):
        """(ms_total, [ms per launch, in kernel_names() order]) measured with HIP events on the launch stream."""
        total = C.c_float()
        k = (C.c_float * 3)()
        _abi.check(self._L.ocean_time_frames(self._h, t0, dt, warmup, frames, C.byref(total),
                                             k if per_kernel else None), "ocean_time_frames")
        return total.value, [k[0], k[1], k[2]] if per_kernel else None

    def kernel_names(self):
        return [self._L.ocean_kernel_name(self._h, i).decode() for i in range(3)]

    def last_launch(self):
        """What the most recent frame launched: three dicts (ocean_launch_info) in kernel_names() order."""
        out = []
        for i in range(3):
            li = _abi.LaunchInfo()
            _abi.check(self._L.ocean_last_launch(self._h, i, C.byref(li)), "ocean_last_launch")
            out.append({k: int(getattr(li, k)) for k, _ in _abi.LaunchInfo._fields_})
        return out

    @property
    def algorithmic_bytes_per_texel(self) -> int:
        return int(self._L.ocean_algorithmic_bytes_per_texel(self._h))

    def algorithmic_bytes_per_launch(self):
        """Bytes per texel of each of the three launches, in kernel_names() order (sums to algorithmic_bytes_per_texel)."""
        return [int(self._L.ocean_algorithmic_bytes_per_launch(self._h, i)) for i in range(3)]


def comm_unique_id() -> bytes:
    """128-byte RCCL id created by one rank and handed to every rank's comm_init."""
    buf = C.create_string_buffer(_abi.OCEAN_COMM_ID_BYTES)
    _abi.check(_abi.lib().ocean_comm_unique_id(buf), "ocean_comm_unique_id")
    return buf.raw


def host_register(arr: np.ndarray):
    _abi.check(_abi.lib().ocean_host_register(arr.ctypes.data_as(C.c_void_p), arr.nbytes), "ocean_host_register")


def host_unregister(arr: np.ndarray):
    _abi.check(_abi.lib().ocean_host_unregister(arr.ctypes.data_as(C.c_void_p)), "ocean_host_unregister")


class WSTessendorf:
    """Drop-in mirror of the reference class (WSTessendorf.h:58-122).

    Like the reference: setters other than SetLambda take effect at the next
    Prepare(); SetTileSize silently ignores a non power of two
    (WSTessendorf.cpp:459-468); nothing raises for "wrong order" except calling
    ComputeWaves before Prepare, which the reference would crash on.
    Prepare() with no seed draws a fresh one, as the reference re-randomises on
    every Prepare (WSTessendorf.cpp:87-103 + core/Application.cpp:21).
    """

    s_kDefaultTileSize = 512
    s_kDefaultTileLength = 1000.0
    s_kDefaultWindDir = (1.0, 1.0)
    s_kDefaultWindSpeed = 30.0
    s_kDefaultAnimPeriod = 200.0
    s_kDefaultPhillipsConst = 3e-7
    s_kDefaultPhillipsDamping = 0.1

    def __init__(self, tileSize: int = 512, tileLength: float = 1000.0, device: int = 0):
        if not _is_pow2(tileSize):
            tileSize = self.s_kDefaultTileSize
        self._b = OceanBatch(tileSize, 1, device)
        self._b.set_params(tile_length=tileLength)
        self._disp = None
        self._nrm = None
        self._min = -1.0   # WSTessendorf.h:227-228
        self._max = 1.0
        self._seed_ctr = 0
        self._back = None      # pinned back pair of ComputeWavesAsync
        self._pending = None   # (A, min, max) of a ComputeWavesAsync whose copy has not been waited for

    # -- Prepare / ComputeWaves (WSTessendorf.cpp:36-58, 284-455) -------------------
    def Prepare(self, seed: int | None = None, xi: np.ndarray | None = None):
        if seed is None:
            import time
            self._seed_ctr += 1
            seed = (time.time_ns() ^ (self._seed_ctr * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
        if self._pending is not None:
            self.Wait()
        self._b.prepare(seed, xi)
        n = self._b.tile_size
        self._disp = np.zeros((n, n, 4), dtype=np.float32)            # .cpp:48-51
        self._nrm = np.zeros((n, n, 4), dtype=np.float32)
        self._nrm[..., 1] = 1.0                                        # .cpp:53-54

    def ComputeWaves(self, time: float) -> float:
        if self._pending is not None:
            self.Wait()
        amp = float(self._b.compute_waves(time)[0])
        d, q = self._b.read_maps(0, 1)
        self._disp, self._nrm = d[0], q[0]
        _, self._min, self._max = self._b.heights(0)
        return amp

    # -- opt-in non-blocking pair (include/WSTessendorf.hpp: ComputeWavesAsync / Wait; the reference's DOUBLE_BUFFERED idea,
    #    WaterSurfaceMesh.h:26-34, on the synthesis side) ---------------------------------------------------------------------
    def SelectFastestQueue(self, framesPerQueue: int = 50):
        """Opt-in, once after Prepare(): the model's work on the fastest hardware queue of the process (include/WSTessendorf.hpp)."""
        self.Wait()
        return self._b.select_streams(framesPerQueue)

    def ComputeWavesAsync(self, time: float) -> float:
        """Enqueue the frame and the DMA of both maps into a back pair of pinned arrays; return A as soon as the frame's kernels
        are done (ocean_wait_frame).  GetDisplacements() / GetNormals() keep returning the previous frame until Wait()."""
        if self._pending is not None:
            self.Wait()
        n = self._b.tile_size
        if self._back is None or self._back[0].shape[1] != n:
            self._release_back()
            self._back = (np.zeros((1, n, n, 4), np.float32), np.zeros((1, n, n, 4), np.float32))
            for a in self._back:
                host_register(a)
            self._b.set_frame_tracking(True)
        self._b.compute_waves_async(time)
        self._b.read_maps_async(self._back[0], self._back[1])
        amp = float(self._b.wait_frame()[0])
        self._pending = self._b.heights(0)
        return amp

    def Wait(self):
        if self._pending is None:
            return
        self._b.synchronize()
        self._disp, self._nrm = self._back[0][0].copy(), self._back[1][0].copy()
        _, self._min, self._max = self._pending
        self._pending = None

    def _release_back(self):
        if getattr(self, "_back", None) is not None:
            for a in self._back:
                try:
                    host_unregister(a)
                except Exception:
                    pass
        self._back = None

    def __del__(self):
        try:
            if self._pending is not None:
                self._b.synchronize()
            self._release_back()
        except Exception:
            pass

    # -- getters (WSTessendorf.h:82-107) ----------------------------------------------
    def GetTileSize(self): return self._b.tile_size
    def GetTileLength(self): return self._b.get_params().tile_length

    def GetWindDir(self):
        p = self._b.get_params()
        inv = 1.0 / math.sqrt(p.wind_dir_x ** 2 + p.wind_dir_y ** 2)
        return (p.wind_dir_x * inv, p.wind_dir_y * inv)

    def GetWindSpeed(self): return max(1e-4, self._b.get_params().wind_speed)
    def GetAnimationPeriod(self): return self._b.get_params().anim_period
    def GetPhillipsConst(self): return self._b.get_params().phillips_const
    def GetDamping(self): return self._b.get_params().damping
    def GetDisplacementLambda(self): return self._b.get_params().lambda_
    def GetMinHeight(self): return self._min
    def GetMaxHeight(self): return self._max
    def GetDisplacementCount(self): return 0 if self._disp is None else self._disp.shape[0] * self._disp.shape[1]
    def GetDisplacements(self): return self._disp
    def GetNormalCount(self): return 0 if self._nrm is None else self._nrm.shape[0] * self._nrm.shape[1]
    def GetNormals(self): return self._nrm

    # -- setters (WSTessendorf.cpp:459-505) ----------------------------------------------
    def SetTileSize(self, size: int):
        if not _is_pow2(size):
            return                      # .cpp:463-467: ignored
        if self._pending is not None:
            self.Wait()
        self._b.set_tile_size(size)

    def SetTileLength(self, length: float): self._b.set_params(tile_length=length)
    def SetWindDirection(self, w): self._b.set_params(wind_dir_x=float(w[0]), wind_dir_y=float(w[1]))
    def SetWindSpeed(self, v: float): self._b.set_params(wind_speed=max(1e-4, v))
    def SetAnimationPeriod(self, T: float): self._b.set_params(anim_period=T)
    def SetPhillipsConst(self, A: float): self._b.set_params(phillips_const=A)
    def SetLambda(self, lam: float): self._b.set_lambda(lam)
    def SetDamping(self, damping: float): self._b.set_params(damping=damping)
