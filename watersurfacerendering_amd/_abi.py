"""ctypes binding of include/ocean.h (libocean_hip.so).

The product path is HIP only: importing this module never touches oracle/, and
every call raises OceanError when the extension or a gfx950 device is missing.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
_BUILT_LIB = os.path.join(_PKG, "libocean_hip.so")
LIB_PATH = _BUILT_LIB      # the one library this package loads (developer A/B scripts under tools/ re-point it themselves: tools/devlib.py)
CSRC = os.path.join(_PKG, "csrc")

OCEAN_OK = 0
OCEAN_E_INVALID = -1
OCEAN_E_NO_DEVICE = -2
OCEAN_E_HIP = -3
OCEAN_E_NOT_READY = -4
OCEAN_E_NOMEM = -5
OCEAN_E_UNSUPPORTED = -6
OCEAN_E_COMM = -7
OCEAN_COMM_ID_BYTES = 128
OCEAN_ALL_TILES = 0xFFFFFFFF
OCEAN_MODE_FULL7, OCEAN_MODE_CHOPPY5, OCEAN_MODE_HEIGHT1, OCEAN_MODE_JACOBIAN = 0, 1, 2, 3

#: every symbol the library's three headers declare (tests check the .so exports each one, and that each list equals its header's)
SYMBOLS_BOUNDARY = [        # include/ocean.h: the drop-in boundary
    "ocean_default_params", "ocean_strerror", "ocean_abi_version", "ocean_build_id", "ocean_last_hip_error", "ocean_fault_recoveries",
    "ocean_create", "ocean_destroy", "ocean_set_params", "ocean_get_params", "ocean_set_lambda",
    "ocean_set_tile_size", "ocean_tile_size", "ocean_tiles", "ocean_prepare",
    "ocean_compute_waves", "ocean_compute_waves_async", "ocean_wait_frame", "ocean_set_frame_tracking", "ocean_set_time_offsets", "ocean_synchronize",
    "ocean_get_heights", "ocean_read_maps", "ocean_compute_waves_read", "ocean_host_register", "ocean_host_unregister",
    "ocean_read_maps_async", "ocean_staging_map_offset", "ocean_read_maps_staging", "ocean_device_maps", "ocean_set_external_readers", "ocean_export_maps", "ocean_bind_output", "ocean_bind_output_dmabuf",
    "ocean_comm_unique_id", "ocean_comm_init", "ocean_comm_destroy", "ocean_comm_count", "ocean_gather_maps", "ocean_gather_maps_f16", "ocean_last_rccl_error",
    "ocean_set_mode", "ocean_set_dispersion", "ocean_set_spectrum_precision", "ocean_set_intermediate_precision", "ocean_set_pipeline_depth", "ocean_stream", "ocean_set_stream",
]
SYMBOLS_CONSUMERS = [       # include/ocean_consumers.h: SURVEY.md 8f ranks 3-4
    "ocean_displace_grid", "ocean_displace_grid_cascades", "ocean_read_grid", "ocean_device_grid",
    "ocean_mip_texels", "ocean_build_mips", "ocean_read_mips", "ocean_device_mips",
]
SYMBOLS_DEV = [             # include/ocean_dev.h: tests, bench.py, tools/
    "ocean_read_spectrum", "ocean_read_xi",
    "ocean_select_streams", "ocean_set_start_ramp", "ocean_set_merged_xpass", "ocean_set_placement_search", "ocean_placement_report", "ocean_time_frames", "ocean_kernel_name", "ocean_last_launch", "ocean_algorithmic_bytes_per_texel", "ocean_algorithmic_bytes_per_launch",
]
SYMBOLS = SYMBOLS_BOUNDARY + SYMBOLS_CONSUMERS + SYMBOLS_DEV
HEADERS = {"ocean.h": SYMBOLS_BOUNDARY, "ocean_consumers.h": SYMBOLS_CONSUMERS, "ocean_dev.h": SYMBOLS_DEV}

OCEAN_LAUNCH_NT_MAPS, OCEAN_LAUNCH_NT_INTER, OCEAN_LAUNCH_HALF_INTER, OCEAN_LAUNCH_JACOBIAN = 1, 2, 4, 8
OCEAN_LAUNCH_FP16_SPECTRUM, OCEAN_LAUNCH_FP32_DISPERSION, OCEAN_LAUNCH_SPLIT_LAST_ROUND, OCEAN_LAUNCH_SINGLE_TRANSFORM = 16, 32, 64, 128
OCEAN_LAUNCH_STAGGERED_START = 256
OCEAN_LAUNCH_SPLIT_ORDER = 512      # developer builds only
OCEAN_LAUNCH_MERGED_X = 1024
OCEAN_LAUNCH_WT_INTER = 2048
OCEAN_LAUNCH_ONE_LAUNCH = 4096


class OceanError(RuntimeError):
    def __init__(self, code: int, what: str):
        super().__init__(f"{what}: {strerror(code)} (code {code}, hip {last_hip_error()}, rccl {last_rccl_error()})")
        self.code = code


class Params(C.Structure):
    """struct ocean_params (include/ocean.h)."""
    _fields_ = [("tile_length", C.c_float), ("wind_dir_x", C.c_float), ("wind_dir_y", C.c_float),
                ("wind_speed", C.c_float), ("anim_period", C.c_float), ("phillips_const", C.c_float),
                ("damping", C.c_float), ("lambda_", C.c_float)]


class LaunchInfo(C.Structure):
    """struct ocean_launch_info (include/ocean.h)."""
    _fields_ = [("tile_size", C.c_uint32), ("grid_x", C.c_uint32), ("grid_y", C.c_uint32), ("block", C.c_uint32),
                ("lds_bytes", C.c_uint32), ("flags", C.c_uint32), ("per_workgroup", C.c_uint32), ("mode", C.c_uint32)]


last_build = ""      # what the most recent build() did, for the caller to log

_ID_MARK = b"OCEAN_BUILD_ID:"


def source_build_id(defs: str = "") -> str:
    """Content hash of the library's sources, computed exactly as csrc/Makefile does (BUILD_ID): SHA-256 over Makefile, *.h, *.hip of
    csrc/ in byte order of their names, then include/ocean.h, ocean_consumers.h, ocean_dev.h, then the build's extra definitions; first 16 hex digits."""
    import hashlib
    h = hashlib.sha256()
    names = sorted(f for f in os.listdir(CSRC) if f == "Makefile" or (f.endswith((".h", ".hip")) and os.path.isfile(os.path.join(CSRC, f))))
    for f in names:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    for f in ("ocean.h", "ocean_consumers.h", "ocean_dev.h"):
        h.update(open(os.path.join(os.path.dirname(_PKG), "include", f), "rb").read())
    h.update(defs.encode())
    return h.hexdigest()[:16]


def library_build_id(path: str = None) -> str | None:
    """The id embedded in a built library (ocean_build_id()), read from the file's bytes: no dlopen, so a stale library is never loaded
    just to find out that it is stale.  None: no such file, or a library from before the id existed."""
    path = path or _BUILT_LIB
    try:
        data = open(path, "rb").read()
    except OSError:
        return None
    i = data.find(_ID_MARK)
    if i < 0:
        return None
    j = data.find(b"\0", i)
    return data[i + len(_ID_MARK):j].decode("ascii", "replace")


def build(force: bool = False) -> str:
    """Compile libocean_hip.so for gfx950 with hipcc (cross-compiles without a GPU) unless the library beside the sources was built
    from exactly these sources: the decision compares CONTENT hashes (ocean_build_id() against source_build_id()), never file times --
    after an rsync or a checkout the times mean nothing.  force (or OCEAN_FORCE_REBUILD=1) rebuilds every translation unit."""
    force = force or os.environ.get("OCEAN_FORCE_REBUILD", "") not in ("", "0")
    want, have = source_build_id(), library_build_id()
    global last_build
    if not force and have == want:
        last_build = f"libocean_hip.so carries build id {have} = the hash of the sources beside it (no compile; OCEAN_FORCE_REBUILD=1 rebuilds from scratch)"
        return _BUILT_LIB
    why = "forced" if force else ("no library yet" if have is None and not os.path.exists(_BUILT_LIB) else f"library build id {have} != source hash {want}")
    subprocess.run(["make", "-C", CSRC] + (["-B"] if force else []), check=True, stdout=subprocess.DEVNULL)
    if library_build_id() != want:
        # make trusted file times that lie (objects newer than sources they were not built from): everything from scratch
        subprocess.run(["make", "-C", CSRC, "-B"], check=True, stdout=subprocess.DEVNULL)
        why += "; incremental make left a stale library, rebuilt from scratch"
    if library_build_id() != want:
        raise RuntimeError(f"libocean_hip.so has build id {library_build_id()} after a full rebuild, sources hash to {want}")
    last_build = f"rebuilt libocean_hip.so ({why}); build id {want}"
    return _BUILT_LIB


_lib = None


def lib() -> C.CDLL:
    """Load the extension.  Fails loudly if it has not been built: no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the ocean synthesis path.")
    # PyTorch's ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  If this library
    # (linked against /opt/rocm) is loaded first and torch afterwards, the process ends up with
    # two HSA runtimes and the second one cannot open the GPU.  Loading torch's copy first makes
    # the dynamic linker satisfy our libamdhip64.so.7 dependency with it: one runtime either way.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    P, u32, u64, f32, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_int
    FP = C.POINTER(C.c_float)
    sig = {
        "ocean_default_params": (None, [C.POINTER(Params)]),
        "ocean_strerror": (C.c_char_p, [i32]),
        "ocean_abi_version": (i32, []),
        "ocean_build_id": (C.c_char_p, []),
        "ocean_last_hip_error": (i32, []),
        "ocean_fault_recoveries": (C.c_uint, [P]),
        "ocean_set_placement_search": (i32, [P, i32]),
        "ocean_placement_report": (i32, [P, C.POINTER(i32), FP, FP]),
        "ocean_compute_waves_read": (i32, [P, f32, FP, P, P]),
        "ocean_create": (i32, [C.POINTER(P), u32, u32, i32]),
        "ocean_destroy": (None, [P]),
        "ocean_set_params": (i32, [P, u32, C.POINTER(Params)]),
        "ocean_get_params": (i32, [P, u32, C.POINTER(Params)]),
        "ocean_set_lambda": (i32, [P, u32, f32]),
        "ocean_set_tile_size": (i32, [P, u32]),
        "ocean_tile_size": (u32, [P]),
        "ocean_tiles": (u32, [P]),
        "ocean_prepare": (i32, [P, u64, C.c_void_p]),
        "ocean_compute_waves": (i32, [P, f32, FP]),
        "ocean_compute_waves_async": (i32, [P, f32]),
        "ocean_wait_frame": (i32, [P, FP]),
        "ocean_set_frame_tracking": (i32, [P, i32]),
        "ocean_set_time_offsets": (i32, [P, C.c_void_p]),
        "ocean_synchronize": (i32, [P]),
        "ocean_get_heights": (i32, [P, u32, FP, FP, FP]),
        "ocean_read_maps": (i32, [P, u32, u32, C.c_void_p, C.c_void_p]),
        "ocean_host_register": (i32, [C.c_void_p, C.c_size_t]),
        "ocean_host_unregister": (i32, [C.c_void_p]),
        "ocean_read_maps_async": (i32, [P, u32, u32, C.c_void_p, C.c_void_p]),
        "ocean_staging_map_offset": (C.c_size_t, [C.c_size_t, C.c_size_t]),
        "ocean_read_maps_staging": (i32, [P, u32, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t)]),
        "ocean_comm_unique_id": (i32, [C.c_void_p]),
        "ocean_comm_init": (i32, [P, i32, i32, C.c_void_p]),
        "ocean_comm_destroy": (i32, [P]),
        "ocean_comm_count": (i32, [P, C.POINTER(i32), C.POINTER(i32)]),
        "ocean_gather_maps": (i32, [P, i32, C.c_void_p, C.c_void_p]),
        "ocean_gather_maps_f16": (i32, [P, i32, C.c_void_p, C.c_void_p]),
        "ocean_last_rccl_error": (i32, []),
        "ocean_device_maps": (i32, [P, C.POINTER(P), C.POINTER(P)]),
        "ocean_bind_output": (i32, [P, P, P]),
        "ocean_set_external_readers": (i32, [P, i32]),
        "ocean_bind_output_dmabuf": (i32, [P, i32, C.c_size_t, C.c_size_t, C.c_size_t]),
        "ocean_export_maps": (i32, [P, C.POINTER(i32), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(i32)]),
        "ocean_displace_grid": (i32, [P, u32, u32, f32, f32, f32]),
        "ocean_displace_grid_cascades": (i32, [P, u32, u32, u32, f32, FP, f32]),
        "ocean_read_grid": (i32, [P, C.c_void_p, C.c_void_p]),
        "ocean_device_grid": (i32, [P, C.POINTER(P), C.POINTER(P), C.POINTER(u32)]),
        "ocean_mip_texels": (C.c_size_t, [u32]),
        "ocean_build_mips": (i32, [P, u32]),
        "ocean_read_mips": (i32, [P, C.c_void_p, C.c_void_p]),
        "ocean_device_mips": (i32, [P, C.POINTER(P), C.POINTER(P), C.POINTER(u32)]),
        "ocean_set_mode": (i32, [P, i32]),
        "ocean_set_dispersion": (i32, [P, i32, f32]),
        "ocean_set_spectrum_precision": (i32, [P, i32]),
        "ocean_set_intermediate_precision": (i32, [P, i32]),
        "ocean_set_pipeline_depth": (i32, [P, i32]),
        "ocean_stream": (P, [P]),
        "ocean_set_stream": (i32, [P, P]),
        "ocean_read_spectrum": (i32, [P, u32, C.c_void_p, C.c_void_p]),
        "ocean_read_xi": (i32, [P, u32, C.c_void_p]),
        "ocean_select_streams": (i32, [P, u32, FP]),
        "ocean_set_start_ramp": (i32, [P, i32]),
        "ocean_set_merged_xpass": (i32, [P, i32]),
        "ocean_time_frames": (i32, [P, f32, f32, i32, i32, FP, FP]),
        "ocean_kernel_name": (C.c_char_p, [P, i32]),
        "ocean_last_launch": (i32, [P, i32, C.POINTER(LaunchInfo)]),
        "ocean_algorithmic_bytes_per_texel": (i32, [P]),
        "ocean_algorithmic_bytes_per_launch": (i32, [P, i32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def strerror(code: int) -> str:
    try:
        return lib().ocean_strerror(code).decode()
    except Exception:  # pragma: no cover
        return "?"


def last_hip_error() -> int:
    try:
        return lib().ocean_last_hip_error()
    except Exception:  # pragma: no cover
        return -1


def last_rccl_error() -> int:
    try:
        return lib().ocean_last_rccl_error()
    except Exception:  # pragma: no cover
        return -1


def check(code: int, what: str) -> None:
    if code != OCEAN_OK:
        raise OceanError(code, what)
