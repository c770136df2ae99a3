"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every
symbol include/ocean.h declares, and its argument checking / error paths work
without a GPU.  No compute is launched here."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def abi():
    from watersurfacerendering_amd import _abi
    _abi.build()
    return _abi


HEADERS = ("ocean.h", "ocean_consumers.h", "ocean_dev.h")


def header_text(name=None):
    names = HEADERS if name is None else (name,)
    return "\n".join(re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", n)).read(), flags=re.S) for n in names)


def declared_symbols(name=None):
    return sorted(set(re.findall(r"\b(ocean_[a-z0-9_]+)\s*\(", header_text(name))))


def test_header_and_binding_list_agree(abi):
    assert declared_symbols() == sorted(abi.SYMBOLS)
    for name, symbols in abi.HEADERS.items():
        assert declared_symbols(name) == sorted(symbols), name
    assert len(abi.SYMBOLS) == len(set(abi.SYMBOLS))


def test_the_boundary_header_keeps_the_survey_shape(abi):
    """include/ocean.h is the SURVEY.md 8b shape -- lifetime, properties, Prepare, ComputeWaves, read-out, the gather --
    and nothing of the bench / A-B plumbing, which lives in include/ocean_dev.h; at most 50 entry points."""
    boundary = declared_symbols("ocean.h")
    assert len(boundary) <= 50, len(boundary)
    for dev in ("ocean_time_frames", "ocean_kernel_name", "ocean_last_launch", "ocean_select_streams", "ocean_set_start_ramp", "ocean_set_merged_xpass",
                "ocean_algorithmic_bytes_per_texel", "ocean_algorithmic_bytes_per_launch", "ocean_read_spectrum", "ocean_read_xi"):
        assert dev not in boundary and dev in declared_symbols("ocean_dev.h"), dev
    for must in ("ocean_create", "ocean_destroy", "ocean_set_params", "ocean_get_params", "ocean_prepare", "ocean_compute_waves", "ocean_read_maps",
                 "ocean_device_maps", "ocean_get_heights", "ocean_gather_maps", "ocean_read_maps_staging", "ocean_export_maps"):
        assert must in boundary, must
    assert "OCEAN_LAUNCH_" not in header_text("ocean.h")            # the launch-variant flags are introspection: ocean_dev.h


def test_header_enumerators_match_the_binding(abi):
    """Every OCEAN_* enumerator of include/ocean.h that the ctypes binding names has the header's value, and the binding names all of the
    launch flags, modes and error codes (a flag added on one side only -- OCEAN_LAUNCH_STAGGERED_START was the last one -- fails here)."""
    txt = header_text()
    enums = {k: int(v) for k, v in re.findall(r"\b(OCEAN_[A-Z0-9_]+)\s*=\s*(-?\d+)", txt)}
    assert len(enums) > 20
    for name, value in enums.items():
        if name.startswith(("OCEAN_LAUNCH_", "OCEAN_MODE_", "OCEAN_E_")) or name == "OCEAN_OK":
            assert hasattr(abi, name), name
        if hasattr(abi, name):
            assert getattr(abi, name) == value, (name, getattr(abi, name), value)


def test_library_exports_every_declared_symbol(abi):
    L = abi.lib()
    for s in declared_symbols():
        assert hasattr(L, s), s
    out = subprocess.run(["nm", "-D", "--defined-only", abi.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (ocean_[a-z0-9_]+)", out))
    assert set(declared_symbols()) <= exported


def test_abi_version_defaults_and_strerror(abi):
    L = abi.lib()
    assert L.ocean_abi_version() == 5
    p = abi.Params()
    L.ocean_default_params(C.byref(p))
    # WSTessendorf.h:36-43,181
    assert (p.tile_length, p.wind_dir_x, p.wind_dir_y, p.wind_speed, p.anim_period) == (1000.0, 1.0, 1.0, 30.0, 200.0)
    assert p.phillips_const == pytest.approx(3e-7) and p.damping == pytest.approx(0.1) and p.lambda_ == -1.0
    for code in range(0, -8, -1):
        assert L.ocean_strerror(code)
    assert b"power of two" in L.ocean_strerror(abi.OCEAN_E_UNSUPPORTED)


def test_argument_checking_without_device(abi):
    L = abi.lib()
    h = C.c_void_p()
    assert L.ocean_create(None, 512, 1, 0) == abi.OCEAN_E_INVALID
    assert L.ocean_create(C.byref(h), 500, 1, 0) == abi.OCEAN_E_INVALID          # not a power of two
    assert L.ocean_create(C.byref(h), 8, 1, 0) == abi.OCEAN_E_UNSUPPORTED
    assert L.ocean_create(C.byref(h), 8192, 1, 0) == abi.OCEAN_E_UNSUPPORTED
    assert L.ocean_create(C.byref(h), 512, 0, 0) == abi.OCEAN_E_INVALID
    assert L.ocean_create(C.byref(h), 16, 65536, 0) == abi.OCEAN_E_UNSUPPORTED    # tiles index blockIdx.y
    assert L.ocean_displace_grid(None, 0, 16, 1.0, 1.0, -1.0) == abi.OCEAN_E_INVALID
    assert L.ocean_read_grid(None, None, None) == abi.OCEAN_E_INVALID
    assert L.ocean_displace_grid_cascades(None, 0, 1, 16, 1.0, None, -1.0) == abi.OCEAN_E_INVALID
    assert L.ocean_compute_waves(None, 0.0, None) == abi.OCEAN_E_INVALID
    assert L.ocean_prepare(None, 0, None) == abi.OCEAN_E_INVALID
    assert L.ocean_tile_size(None) == 0
    assert L.ocean_set_mode(None, 0) == abi.OCEAN_E_INVALID
    assert L.ocean_set_pipeline_depth(None, 2) == abi.OCEAN_E_INVALID
    assert L.ocean_set_spectrum_precision(None, 16) == abi.OCEAN_E_INVALID
    assert L.ocean_host_register(None, 16) == abi.OCEAN_E_INVALID
    assert L.ocean_read_maps_async(None, 0, 1, None, None) == abi.OCEAN_E_INVALID
    assert L.ocean_kernel_name(None, 0) is None
    # staging layout of WaterSurfaceMesh.cpp:19-22,721-724: maps at align16(vertices + indices)
    assert L.ocean_staging_map_offset(0, 0) == 0 and L.ocean_staging_map_offset(1, 0) == 16
    assert L.ocean_staging_map_offset(263169 * 32, 1572864 * 4) == (263169 * 32 + 1572864 * 4 + 15) // 16 * 16
    assert L.ocean_read_maps_staging(None, 0, None, 0, 0, None) == abi.OCEAN_E_INVALID
    # RCCL gather entry points: argument checks need no device (and no librccl)
    assert L.ocean_comm_unique_id(None) == abi.OCEAN_E_INVALID
    assert L.ocean_comm_init(None, 1, 0, None) == abi.OCEAN_E_INVALID
    assert L.ocean_comm_destroy(None) == abi.OCEAN_E_INVALID
    assert L.ocean_gather_maps(None, 0, None, None) == abi.OCEAN_E_INVALID
    assert L.ocean_gather_maps_f16(None, 0, None, None) == abi.OCEAN_E_INVALID
    assert L.ocean_set_external_readers(None, 0) == abi.OCEAN_E_INVALID
    assert L.ocean_set_start_ramp(None, 0) == abi.OCEAN_E_INVALID
    L.ocean_destroy(None)


def test_build_is_content_addressed(abi, tmp_path, monkeypatch):
    """build() decides by CONTENT (the id embedded in the library against the hash of the sources beside it), never by file
    times -- after an rsync or a checkout they mean nothing.  The id is read from the file's bytes (no dlopen of a possibly stale
    library); a stale library with a NEWER time stamp is found stale, a fresh one with an OLDER time stamp is not rebuilt."""
    want = abi.source_build_id()
    assert len(want) == 16 and abi.library_build_id() == want == abi.lib().ocean_build_id().decode()
    # the Makefile computes the same id (what the compile embeds)
    out = subprocess.run(["make", "-s", "-C", abi.CSRC, "--eval", "print-id: ; @echo $(BUILD_ID)", "print-id"], capture_output=True, text=True)
    assert out.stdout.strip() == want, (out.stdout, out.stderr)
    # a library built from other sources, however new its time stamp: stale (make is not run here: the decision is what is tested)
    calls = []
    monkeypatch.setattr(abi.subprocess, "run", lambda cmd, **kw: calls.append(cmd) or subprocess.CompletedProcess(cmd, 0))
    fake = tmp_path / "libocean_hip.so"
    data = open(abi.LIB_PATH, "rb").read()
    i = data.find(b"OCEAN_BUILD_ID:" + want.encode())
    assert i >= 0
    fake.write_bytes(data[:i] + b"OCEAN_BUILD_ID:" + b"0" * 16 + data[i + 31:])
    os.utime(fake, (2e9, 2e9))                                 # far in the future: newer than every source
    monkeypatch.setattr(abi, "_BUILT_LIB", str(fake))
    assert abi.library_build_id() == "0" * 16
    with pytest.raises(RuntimeError):                          # (the fake make changes nothing, so build() ends by saying so)
        abi.build()
    assert calls and calls[0][0] == "make" and "-B" in calls[-1]      # incremental first, then from scratch
    # the real library with a time stamp OLDER than every source: up to date, no compile
    calls.clear()
    old = tmp_path / "old.so"
    old.write_bytes(data)
    os.utime(old, (1.0, 1.0))
    monkeypatch.setattr(abi, "_BUILT_LIB", str(old))
    abi.build()
    assert not calls and "no compile" in abi.last_build and want in abi.last_build


def test_no_cpu_fallback_when_no_gpu(abi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = abi.lib()
    h = C.c_void_p()
    assert L.ocean_create(C.byref(h), 512, 1, 0) == abi.OCEAN_E_NO_DEVICE
    assert not h.value
    import watersurfacerendering_amd as W
    with pytest.raises(W.OceanError):
        W.WSTessendorf(64)


def test_product_path_never_imports_oracle():
    """The package must not reference oracle/ (a product path through the oracle voids parity)."""
    pkg = os.path.join(ROOT, "watersurfacerendering_amd")
    bad = re.compile(r"(^\s*(import|from)\s+oracle\b|libocean_oracle|ocean_oracle|oracle_[a-z0-9_]+\s*\()", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                assert not bad.search(open(os.path.join(dirpath, f)).read()), (dirpath, f)
    for f in ("ocean.h", "WSTessendorf.hpp"):
        assert not bad.search(open(os.path.join(ROOT, "include", f)).read())
    code = "import sys; import watersurfacerendering_amd; assert not any(m.startswith('oracle') for m in sys.modules)"
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)


REFERENCE_METHODS = [  # /root/reference/src/scene/WSTessendorf.h:58-122
    "Prepare", "ComputeWaves", "GetTileSize", "GetTileLength", "GetWindDir", "GetWindSpeed", "GetAnimationPeriod",
    "GetPhillipsConst", "GetDamping", "GetDisplacementLambda", "GetMinHeight", "GetMaxHeight",
    "GetDisplacementCount", "GetDisplacements", "GetNormalCount", "GetNormals", "SetTileSize", "SetTileLength",
    "SetWindDirection", "SetWindSpeed", "SetAnimationPeriod", "SetPhillipsConst", "SetLambda", "SetDamping",
]


def test_reference_method_list_is_the_reference_headers():
    """Where the reference tree is present (the build container; never the GPU box) the list above is checked against the
    public section of its header: every public method of WSTessendorf, none missing, none invented."""
    hdr_path = "/root/reference/src/scene/WSTessendorf.h"
    if not os.path.exists(hdr_path):
        pytest.skip("reference tree not present")
    text = open(hdr_path).read()
    public = text[text.index("public:"):text.index("private:")]
    public = re.sub(r"/\*.*?\*/", "", public, flags=re.S)
    public = re.sub(r"//[^\n]*", "", public)
    names = set(re.findall(r"\b((?:Get|Set)[A-Za-z]+|Prepare|ComputeWaves)\s*\(", public))
    assert names == set(REFERENCE_METHODS), (sorted(names - set(REFERENCE_METHODS)), sorted(set(REFERENCE_METHODS) - names))


def test_default_parameters_are_the_reference_headers(abi):
    """ocean_default_params against the s_kDefault* constants and the m_Lambda initialiser parsed from the reference header
    (build container only)."""
    hdr_path = "/root/reference/src/scene/WSTessendorf.h"
    if not os.path.exists(hdr_path):
        pytest.skip("reference tree not present")
    text = open(hdr_path).read()

    def const(name):
        m = re.search(r"\b%s\s*\{([^}]*)\}" % name, text)
        assert m, name
        return [float(x.strip().rstrip("f")) for x in m.group(1).split(",")]

    p = abi.Params()
    abi.lib().ocean_default_params(C.byref(p))
    f32 = lambda x: C.c_float(x).value
    assert [p.tile_length] == [f32(v) for v in const("s_kDefaultTileLength")]
    assert [p.wind_dir_x, p.wind_dir_y] == [f32(v) for v in const("s_kDefaultWindDir")]
    assert [p.wind_speed] == [f32(v) for v in const("s_kDefaultWindSpeed")]
    assert [p.anim_period] == [f32(v) for v in const("s_kDefaultAnimPeriod")]
    assert [p.phillips_const] == [f32(v) for v in const("s_kDefaultPhillipsConst")]
    assert [p.damping] == [f32(v) for v in const("s_kDefaultPhillipsDamping")]
    assert [p.lambda_] == [f32(v) for v in const("m_Lambda")]
    import watersurfacerendering_amd as W
    assert [float(W.WSTessendorf.s_kDefaultTileSize)] == const("s_kDefaultTileSize")


def test_python_mirror_has_reference_surface():
    import watersurfacerendering_amd as W
    for m in REFERENCE_METHODS:
        assert callable(getattr(W.WSTessendorf, m)), m
    assert W.WSTessendorf.s_kDefaultTileSize == 512 and W.WSTessendorf.s_kDefaultTileLength == 1000.0


def test_cpp_adaptor_has_reference_surface_and_links(abi, tmp_path):
    hdr = open(os.path.join(ROOT, "include", "WSTessendorf.hpp")).read()
    for m in REFERENCE_METHODS:
        assert re.search(r"\b%s\s*\(" % m, hdr), m
    exe = tmp_path / "adaptor_demo"
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "adaptor_demo.cpp"), "-o", str(exe),
           "-L", os.path.dirname(abi.LIB_PATH), "-locean_hip", "-Wl,-rpath," + os.path.dirname(abi.LIB_PATH),
           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([str(exe), "64"], capture_output=True, text=True)
        assert r.returncode == 3 and "no usable HIP device" in r.stderr   # loud failure, no fallback


def test_abi_header_is_plain_c99(abi, tmp_path):
    """The boundary is a C ABI: include/ocean.h must compile as strict C99 and link from a C program."""
    src = tmp_path / "c_abi.c"
    src.write_text('#include "ocean.h"\n#include "ocean_consumers.h"\n#include "ocean_dev.h"\n#include <stdio.h>\n'
                   'int main(void) { ocean_params p; ocean_t* h = 0; ocean_default_params(&p);\n'
                   '  if (ocean_abi_version() <= 0 || p.tile_length != 1000.0f) return 1;\n'
                   '  if (ocean_create(&h, 500, 1, 0) != OCEAN_E_INVALID) return 2;   /* not a power of two */\n'
                   '  printf("%s\\n", ocean_strerror(OCEAN_E_NO_DEVICE)); return 0; }\n')
    exe = tmp_path / "c_abi"
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", os.path.dirname(abi.LIB_PATH), "-locean_hip", "-Wl,-rpath," + os.path.dirname(abi.LIB_PATH),
                    "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip()


def test_missing_extension_fails_loudly():
    """No silent fallback: without the built .so the package refuses to work."""
    code = ("import watersurfacerendering_amd as W\n"
            "from watersurfacerendering_amd import _abi\n"
            "_abi.LIB_PATH = '/nonexistent/libocean_hip.so'\n"
            "try:\n    W.OceanBatch(64)\nexcept ImportError as e:\n    print('LOUD', e)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert "LOUD" in r.stdout and "no CPU fallback" in r.stdout.replace("There is no CPU fallback", "no CPU fallback")


def test_cpp_gather_host_builds_and_fails_loudly_without_a_gpu(abi, tmp_path):
    """tests/cpp/gather_demo.cpp (the multi-GPU batch mode from a plain C++ host) must compile and link against the C ABI on
    every CPU run; without a GPU it exits non-zero with the library's error text, no silent fallback."""
    exe = tmp_path / "gather_demo"
    lib_dir = os.path.dirname(abi.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                    "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tests", "cpp", "gather_demo.cpp"), "-o", str(exe),
                    "-L", lib_dir, "-locean_hip", "-Wl,-rpath," + lib_dir, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([str(exe), "1", "64", "2", "2"], capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "GATHER_OK" not in r.stdout
        assert "RCCL error" in r.stderr or "no usable HIP device" in r.stderr


def test_cpp_dmabuf_importer_builds(tmp_path):
    """tests/cpp/import_demo.cpp (the importing side of ocean_export_maps, run by the GPU tests) must compile and link on every CPU run; without
    a GPU it fails loudly on its first HIP call."""
    exe = tmp_path / "import_demo"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                    os.path.join(ROOT, "tests", "cpp", "import_demo.cpp"), "-o", str(exe), "-L/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([str(exe), "0", "4096", "0", "2048", "2048", str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "IMPORT_OK" not in r.stdout


def test_export_and_tracking_fail_loudly_without_a_device(abi):
    """The round-3 entry points follow the ABI's error convention on bad arguments (no device needed)."""
    L = abi.lib()
    fd = C.c_int(7)
    assert L.ocean_export_maps(None, C.byref(fd), None, None, None, None) == abi.OCEAN_E_INVALID
    assert L.ocean_wait_frame(None, None) == abi.OCEAN_E_INVALID and L.ocean_set_frame_tracking(None, 1) == abi.OCEAN_E_INVALID
    li = abi.LaunchInfo()
    assert L.ocean_last_launch(None, 0, C.byref(li)) == abi.OCEAN_E_INVALID
    assert L.ocean_comm_count(None, None, None) == abi.OCEAN_E_INVALID
    assert L.ocean_algorithmic_bytes_per_launch(None, 0) == 23 and L.ocean_algorithmic_bytes_per_launch(None, 1) == 28
    assert L.ocean_algorithmic_bytes_per_launch(None, 2) == 22 and L.ocean_algorithmic_bytes_per_launch(None, 3) == 0


def test_shipped_library_reads_no_environment():
    """Every A/B and attribution switch (OCEAN_* variables) lives behind -DOCEAN_DEVELOPER: the shipped library does not import getenv."""
    import shutil
    import subprocess
    from watersurfacerendering_amd import _abi
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--undefined-only", _abi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in out, [l for l in out.splitlines() if "getenv" in l]
