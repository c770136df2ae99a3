"""State and ordering guarantees of the C ABI around the frame path (include/ocean.h), on the GPU:

  * ocean_compute_waves / ocean_wait_frame return from a poll of the frame's completion records (no stream
    synchronisation): amplitudes are those of THIS frame, frames longer than the poll budget take the fallback;
  * a resize after pipelined use leaves no reference to a freed chain (NOT_READY instead of a HIP error);
  * consumer launches (mip chain, vertex stage) of consecutive pipelined frames never write the shared output buffers at once;
  * two contexts driven from two host threads give the bits of serial use (ocean.h: "one context per thread, no hidden globals").

Reference call shape: WaterSurfaceMesh.cpp:145-154 (ComputeWaves per frame, return value -> WSHeightAmp).
"""
import threading

import numpy as np
import pytest

from conftest import isolated, pinned_array      # tests that share memory across processes or exhaust it run in a pytest process of their own

pytestmark = pytest.mark.gpu

SEED = 0x5EED0000


def test_wait_frame_returns_this_frames_amplitude_every_time():
    """500 back-to-back synchronous calls with a different time each: the amplitude returned by the poll is the one a fully
    synchronised read of the same frame gives (a stale completion record would return the previous frame's)."""
    import watersurfacerendering_amd as W
    for n in (64, 512, 2048):
        b = W.OceanBatch(n, 1, 0)
        r = W.OceanBatch(n, 1, 0)
        b.prepare(SEED); r.prepare(SEED)
        ts = [0.37 * j for j in range(500 if n < 2048 else 120)]
        want = []
        for t in ts:
            r.compute_waves_async(t); r.synchronize()
            want.append(r.heights(0))
        for t, w in zip(ts, want):
            a = float(b.compute_waves(t)[0])
            assert a == w[0], (n, t, a, w)
            assert b.heights(0) == w
        # the asynchronous pair: enqueue, enqueue the read-out, wait for the amplitude, then for the copy
        d = pinned_array((1, n, n, 4), np.float32, tag="d"); q = pinned_array((1, n, n, 4), np.float32, tag="q")
        if True:
            for tracking in (False, True):          # untracked: the wait is a stream synchronisation; tracked: a poll of the records
                b.set_frame_tracking(tracking)
                b.compute_waves_async(ts[7 + tracking])
                b.read_maps_async(d, q)
                assert float(b.wait_frame()[0]) == want[7 + tracking][0]
                b.synchronize()
            b.compute_waves_async(ts[7])
            b.read_maps_async(d, q)
            assert float(b.wait_frame()[0]) == want[7][0]
            b.synchronize()
            r.compute_waves(ts[7])
            dr, qr = r.read_maps()
            assert np.array_equal(d, dr) and np.array_equal(q, qr)
        b.close(); r.close()


def test_wait_frame_fallback_for_frames_longer_than_the_poll_budget():
    """8 x 4096^2 per launch is ~2.5 ms of kernels: the poll gives up after 2 ms and takes hipStreamSynchronize; the records are
    there afterwards and every tile's amplitude is right."""
    import watersurfacerendering_amd as W
    n, tiles = 4096, 8
    b = W.OceanBatch(n, tiles, 0)
    b.prepare(SEED)
    amps = b.compute_waves(2.5)
    for i in (0, 3, tiles - 1):
        s = W.OceanBatch(n, 1, 0)
        s.prepare(SEED + i)
        assert float(s.compute_waves(2.5)[0]) == float(amps[i]) == b.heights(i)[0]
        s.close()
    b.close()


def test_pipelined_frames_each_have_their_own_completion_records():
    import watersurfacerendering_amd as W
    n = 256
    b = W.OceanBatch(n, 3, 0)
    b.set_pipeline_depth(4)
    b.set_frame_tracking(True)
    b.prepare(SEED)
    r = W.OceanBatch(n, 3, 0)
    r.prepare(SEED)
    for j in range(23):
        t = 0.21 * j
        if j % 3 == 2:
            got = b.compute_waves(t)              # waits for its own chain only; the others keep running
        else:
            b.compute_waves_async(t)
            got = b.wait_frame() if j % 3 == 1 else None
        if got is not None:
            want = r.compute_waves(t)
            assert np.array_equal(got, want), j
    b.synchronize()
    b.close(); r.close()


def test_resize_after_pipelined_use_forgets_the_old_chains():
    import watersurfacerendering_amd as W
    from watersurfacerendering_amd import _abi
    b = W.OceanBatch(256, 1, 0)
    b.set_pipeline_depth(3)
    b.prepare(SEED)
    b.compute_waves_async(0.1); b.compute_waves_async(0.2)      # the last frame sits on chain 1
    b.build_mips(0)
    b.synchronize()
    b.set_tile_size(128)
    b.prepare(SEED)
    n = 128
    d = np.empty((1, n, n, 4), np.float32); q = np.empty_like(d)
    for call in (lambda: b.read_maps(), lambda: b.read_maps_async(d, q), lambda: b.heights(0), lambda: b.wait_frame(),
                 lambda: b.build_mips(0), lambda: b.displace_grid(0, 16), lambda: b.last_launch()):
        with pytest.raises(W.OceanError) as e:
            call()
        assert e.value.code == _abi.OCEAN_E_NOT_READY
    L = _abi.lib()
    import ctypes as C
    P = C.c_void_p
    dm, qm, lv = P(), P(), C.c_uint32(7)
    assert L.ocean_device_mips(b._h, C.byref(dm), C.byref(qm), C.byref(lv)) == 0 and not dm.value and lv.value == 0
    assert L.ocean_read_mips(b._h, None, None) == _abi.OCEAN_E_NOT_READY
    a = float(b.compute_waves(0.7)[0])
    f = W.OceanBatch(128, 1, 0)
    f.prepare(SEED)
    assert float(f.compute_waves(0.7)[0]) == a
    d1, q1 = b.read_maps(); d2, q2 = f.read_maps()
    assert np.array_equal(d1, d2) and np.array_equal(q1, q2)
    m1, m2 = b.build_mips(0), f.build_mips(0)
    assert all(np.array_equal(x, y) for x, y in zip(m1[0] + m1[1], m2[0] + m2[1]))
    b.close(); f.close()


def test_consumer_launches_of_pipelined_frames_do_not_race_on_their_buffers():
    """At depth 3 the mip chain / vertex stage of frame j runs on chain j % 3's stream; the output buffers are the context's.
    The launches are chained by an event, so what is read back is exactly the last frame's result."""
    import watersurfacerendering_amd as W
    n = 1024
    b = W.OceanBatch(n, 1, 0)
    b.set_pipeline_depth(3)
    b.prepare(SEED)
    r = W.OceanBatch(n, 1, 0)
    r.prepare(SEED)
    L, h = W._abi.lib(), b._h
    for rep in range(6):
        for j in range(5):
            b.compute_waves_async(0.3 * j + rep)
            W._abi.check(L.ocean_build_mips(h, 0), "ocean_build_mips")
            W._abi.check(L.ocean_displace_grid(h, 0, n, 1000.0 / 512.0, 1.0, -1.0), "ocean_displace_grid")
        got_m = b.build_mips(0)                       # one more on the last frame's chain, then read (synchronises)
        got_p, got_n = b.displace_grid(0, n)
        r.compute_waves(0.3 * 4 + rep)
        want_m = r.build_mips(0)
        want_p, want_n = r.displace_grid(0, n)
        assert all(np.array_equal(x, y) for x, y in zip(got_m[0] + got_m[1], want_m[0] + want_m[1])), rep
        assert np.array_equal(got_p, want_p) and np.array_equal(got_n, want_n), rep
    b.close(); r.close()


def test_two_contexts_from_two_host_threads_match_serial_use():
    import watersurfacerendering_amd as W
    cfg = [(512, 2, SEED + 11, 3), (1024, 1, SEED + 29, 1)]
    times = [0.05 * j for j in range(60)]

    def drive(n, tiles, seed, depth, out):
        try:
            b = W.OceanBatch(n, tiles, 0)
            b.set_pipeline_depth(depth)
            b.prepare(seed)
            amps = []
            for j, t in enumerate(times):
                if j % 4 == 3:
                    amps.append(b.compute_waves(t).copy())
                else:
                    b.compute_waves_async(t)
            b.synchronize()
            d, q = b.read_maps()
            out.append((amps, d, q, [b.heights(i) for i in range(tiles)]))
            b.close()
        except Exception as exc:            # surfaces in the main thread
            out.append(exc)

    serial = []
    for c in cfg:
        drive(*c, serial)
    outs = [[] for _ in cfg]
    threads = [threading.Thread(target=drive, args=(*c, o)) for c, o in zip(cfg, outs)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for want, got in zip(serial, outs):
        assert not isinstance(want, Exception), want
        assert len(got) == 1 and not isinstance(got[0], Exception), got
        got = got[0]
        assert all(np.array_equal(x, y) for x, y in zip(want[0], got[0]))
        assert np.array_equal(want[1], got[1]) and np.array_equal(want[2], got[2]) and want[3] == got[3]


@isolated
def test_exported_maps_are_read_by_another_process_through_the_dmabuf(tmp_path):
    """ocean_export_maps (the remainder): the map set as ONE dma-buf a renderer imports instead of the reference's
    staging-buffer round trip (WaterSurfaceMesh.cpp:642-755).  Without Vulkan on the image, the importer is a second process that maps the
    descriptor with hipImportExternalMemory (tests/cpp/import_demo.cpp) and must read, at the offsets the export names, exactly the maps
    this process reads out; a second frame written after the export is visible through the same descriptor (no re-export at depth 1)."""
    import os
    import subprocess
    import watersurfacerendering_amd as W
    from watersurfacerendering_amd import _abi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "import_demo"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                    os.path.join(root, "tests", "cpp", "import_demo.cpp"), "-o", str(exe), "-L/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    n, tiles = 256, 2
    b = W.OceanBatch(n, tiles, 0)
    b.prepare(SEED)
    b.compute_waves(0.75)
    fd, doff, noff, nbytes, mset = b.export_maps()
    try:
        assert fd >= 0 and mset == 0 and doff == 0 and noff == tiles * n * n * 16 and nbytes >= 2 * tiles * n * n * 16 and nbytes % (2 << 20) == 0
        for t in (0.75, 3.5):                     # the second frame is written AFTER the export, into the same memory
            b.compute_waves(t)
            d, q = b.read_maps()                  # (synchronises: the importer reads finished maps)
            out = tmp_path / f"maps_{t}.bin"
            map_bytes = tiles * n * n * 16
            r = subprocess.run([str(exe), str(fd), str(nbytes), str(doff), str(noff), str(map_bytes), str(out)], pass_fds=(fd,),
                               capture_output=True, text=True, timeout=300)
            assert r.returncode == 0 and "IMPORT_OK" in r.stdout, (r.stdout, r.stderr)
            got = np.fromfile(out, dtype=np.float32).reshape(2, tiles, n, n, 4)
            assert np.array_equal(got[0], d) and np.array_equal(got[1], q), t
    finally:
        os.close(fd)
    # caller-bound output is the caller's memory to export
    import torch
    maps = torch.zeros((2, tiles, n, n, 4), dtype=torch.float32, device="cuda:0")
    b.bind_output(maps[0].data_ptr(), maps[1].data_ptr())
    with pytest.raises(W.OceanError) as e:
        b.export_maps()
    assert e.value.code == _abi.OCEAN_E_UNSUPPORTED
    b.bind_output(None, None)
    b.close()


@isolated
def test_importer_sees_every_frame_right_after_the_synchronous_call(tmp_path):
    """The ordering contract of include/ocean.h for readers OUTSIDE the context's streams: a completion record
    is not a memory fence, so once a map set has been exported ocean_compute_waves / ocean_wait_frame synchronise the frame's stream instead
    of polling.  A second process keeps the dma-buf imported (tests/cpp/import_demo.cpp in its loop mode) and reads both maps the moment the
    producer's call returns -- no read-out, no ocean_synchronize in between -- 200 frames, synchronous and tracked asynchronous + wait, each
    bit-exact against a twin context that is read out the usual way."""
    import os
    import subprocess
    import watersurfacerendering_amd as W
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "import_demo"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                    os.path.join(root, "tests", "cpp", "import_demo.cpp"), "-o", str(exe), "-L/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    n, tiles = 512, 2
    b = W.OceanBatch(n, tiles, 0)
    twin = W.OceanBatch(n, tiles, 0)
    b.prepare(SEED + 9); twin.prepare(SEED + 9)
    b.compute_waves(0.0)
    fd, doff, noff, nbytes, _ = b.export_maps()
    map_bytes = tiles * n * n * 16
    child = subprocess.Popen([str(exe), str(fd), str(nbytes), str(doff), str(noff), str(map_bytes), "-"], pass_fds=(fd,),
                             stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1)
    weights = np.arange(1, 2 * map_bytes // 4 + 1, dtype=np.uint64)
    try:
        assert child.stdout.readline().strip() == "IMPORT_READY"
        b.set_frame_tracking(True)
        for j in range(200):
            t = 0.37 * j
            if j % 2 == 0:
                amp = b.compute_waves(t)                       # the reference's call shape
            else:
                b.compute_waves_async(t)                       # tracked frame + wait: the same contract
                amp = b.wait_frame()
            child.stdin.write("r\n"); child.stdin.flush()      # the importer reads NOW
            line = child.stdout.readline().split()
            assert line and line[0] == "SUM", line
            amp_t = twin.compute_waves(t)
            d, q = twin.read_maps()
            words = np.concatenate([d.reshape(-1), q.reshape(-1)]).view(np.uint32).astype(np.uint64)
            want = int((words * weights).sum(dtype=np.uint64))
            assert int(line[1]) == want, f"frame {j}: the importer read something else than the finished maps"
            assert np.array_equal(np.asarray(amp), np.asarray(amp_t))
        child.stdin.write("q\n"); child.stdin.flush()
        assert child.wait(timeout=60) == 0
    finally:
        if child.poll() is None:
            child.kill()
        os.close(fd)
    b.close(); twin.close()


def test_python_mirror_async_pair_matches_the_blocking_call():
    """WSTessendorf.ComputeWavesAsync() / Wait() (the mirror of include/WSTessendorf.hpp's opt-in pair): A at once, the previous frame's maps
    until Wait(), then exactly the blocking call's frame."""
    import watersurfacerendering_amd as W
    ws = W.WSTessendorf(256, 1000.0)
    ws.Prepare(seed=21)
    a0 = ws.ComputeWaves(0.5)
    d0 = ws.GetDisplacements().copy()
    a1 = ws.ComputeWavesAsync(1.5)
    assert a1 != a0 and np.array_equal(ws.GetDisplacements(), d0)          # the front pair is still the previous frame
    ws.Wait()
    d1, q1, mn1, mx1 = ws.GetDisplacements().copy(), ws.GetNormals().copy(), ws.GetMinHeight(), ws.GetMaxHeight()
    assert not np.array_equal(d1, d0)
    assert ws.ComputeWaves(1.5) == a1
    assert np.array_equal(ws.GetDisplacements(), d1) and np.array_equal(ws.GetNormals(), q1)
    assert (ws.GetMinHeight(), ws.GetMaxHeight()) == (mn1, mx1)
    for j in range(5):                                                     # back to back: each call waits for the one before
        ws.ComputeWavesAsync(2.0 + j)
    ws.Wait()
    assert ws.ComputeWaves(6.0) == pytest.approx(ws.ComputeWaves(6.0))


@isolated
def test_maps_written_into_memory_another_process_owns(tmp_path):
    """ocean_bind_output_dmabuf, the other direction of the interop: the renderer owns the memory and hands a dma-buf in.  Here this process
    owns it (a context's exported map set stands in for a VkDeviceMemory), a CHILD process imports the descriptor, binds it as its output and
    synthesises a frame into it; this process then finds exactly that frame's maps in its own memory."""
    import os
    import subprocess
    import sys
    import watersurfacerendering_amd as W
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n, tiles, seed, t = 256, 2, SEED + 77, 2.75
    owner = W.OceanBatch(n, tiles, 0)                  # only its memory is used: [displacement maps | normal maps]
    owner.prepare(1)
    owner.compute_waves(0.0)
    fd, doff, noff, nbytes, _ = owner.export_maps()
    try:
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "workers", "dmabuf_writer.py"), str(fd), str(nbytes), str(doff), str(noff),
                            str(n), str(tiles), str(seed), str(t)], pass_fds=(fd,), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "WRITER_OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
        got_d, got_q = owner.read_maps()               # the owner's map memory, read through the owner
        ref = W.OceanBatch(n, tiles, 0)
        ref.prepare(seed)
        amp = ref.compute_waves(t)
        d, q = ref.read_maps()
        assert np.array_equal(got_d, d) and np.array_equal(got_q, q)
        assert [float(x) for x in r.stdout.split("WRITER_OK")[1].split()] == [float(a) for a in amp]
        ref.close()
    finally:
        os.close(fd)
    # bad arguments are refused before anything is imported
    from watersurfacerendering_amd import _abi
    L = _abi.lib()
    assert L.ocean_bind_output_dmabuf(owner._h, -1, 1 << 20, 0, 0) == _abi.OCEAN_E_INVALID
    assert L.ocean_bind_output_dmabuf(owner._h, 0, 16, 0, 0) == _abi.OCEAN_E_INVALID              # too small for the maps
    assert L.ocean_bind_output_dmabuf(owner._h, 0, nbytes, 0, 64) == _abi.OCEAN_E_INVALID         # overlapping maps
    # offsets so large that offset + map size wraps around must not pass the bounds check
    assert L.ocean_bind_output_dmabuf(owner._h, 0, nbytes, (1 << 64) - 16, 0) == _abi.OCEAN_E_INVALID
    assert L.ocean_bind_output_dmabuf(owner._h, 0, nbytes, 0, (1 << 64) - 32) == _abi.OCEAN_E_INVALID
    owner.close()


@isolated
def test_dmabuf_binding_leaves_the_descriptor_with_the_caller():
    """ocean_bind_output_dmabuf inside ONE process (an allocation of another context stands in for the renderer's), twice with the same
    descriptor: the import holds its own reference, the caller's descriptor stays open and usable through import, re-binding, un-binding and
    ocean_destroy (include/ocean.h says so; CUDA-style ownership transfer would make the caller's close() a double close)."""
    import os
    import watersurfacerendering_amd as W
    n = 256
    owner = W.OceanBatch(n, 1, 0)
    owner.prepare(1)
    owner.compute_waves(0.0)
    fd, doff, noff, nbytes, _ = owner.export_maps()
    ref = W.OceanBatch(n, 1, 0)
    ref.prepare(SEED + 5)
    try:
        b = W.OceanBatch(n, 1, 0)
        b.prepare(SEED + 5)
        for t in (1.0, 2.0):
            b.bind_output_dmabuf(fd, nbytes, doff, noff)
            os.fstat(fd)                                   # still the caller's
            amp = b.compute_waves(t)
            b.synchronize()
            amp_ref = ref.compute_waves(t)
            d, q = owner.read_maps()
            d2, q2 = ref.read_maps()
            assert np.array_equal(amp, amp_ref) and np.array_equal(d, d2) and np.array_equal(q, q2)
            b.bind_output(None, None)                      # ends the binding; the next frame goes to the context's own maps
            os.fstat(fd)
        b.compute_waves(3.0)
        d3, _ = b.read_maps()
        ref.compute_waves(3.0)
        assert np.array_equal(d3, ref.read_maps()[0]) and np.array_equal(owner.read_maps()[0], d)    # and no longer to the owner's
        b.bind_output_dmabuf(fd, nbytes, doff, noff)
        b.close()                                          # destroy with a live binding
        os.fstat(fd)
    finally:
        os.close(fd)
    owner.close(); ref.close()


@isolated
def test_out_of_memory_is_reported_as_such_and_leaves_the_device_usable():
    """A context that cannot fit (60 000 tiles of 4096^2: 8 TB of spectrum alone) is refused with OCEAN_E_NOMEM -- not a generic HIP error, no
    handle, nothing left allocated -- and the next context works and delivers the usual frame (the failed allocation's sticky HIP error is cleared)."""
    import ctypes as C
    import watersurfacerendering_amd as W
    from watersurfacerendering_amd import _abi
    L = _abi.lib()
    ref = W.OceanBatch(256, 1, 0)
    ref.prepare(SEED)
    a0 = ref.compute_waves(1.0)
    d0, q0 = ref.read_maps()
    h = C.c_void_p()
    assert L.ocean_create(C.byref(h), 4096, 60000, 0) == _abi.OCEAN_E_NOMEM
    assert not h.value
    b = W.OceanBatch(256, 1, 0)
    b.prepare(SEED)
    a1 = b.compute_waves(1.0)
    d1, q1 = b.read_maps()
    assert np.array_equal(a0, a1) and np.array_equal(d0, d1) and np.array_equal(q0, q1)
    # a resize that cannot fit: refused the same way, and the context recovers with a size that fits
    big = W.OceanBatch(64, 40000, 0)
    assert L.ocean_set_tile_size(big._h, 4096) == _abi.OCEAN_E_NOMEM
    assert L.ocean_compute_waves_async(big._h, 0.0) == _abi.OCEAN_E_NOT_READY
    big.set_tile_size(16)
    big.prepare(SEED)
    big.compute_waves(0.5)
    big.close(); b.close(); ref.close()


def test_stream_selection_orders_the_queues_and_changes_no_bit():
    """ocean_select_streams: four frame times, fastest first; frames before and after the re-ordering -- serial, pipelined at depth 3, tracked,
    read out -- are the frames of an untouched context bit for bit; the calibration frame is not handed out as a result."""
    import watersurfacerendering_amd as W
    from watersurfacerendering_amd import _abi
    n = 1024
    ref = W.OceanBatch(n, 1, 0)
    ref.prepare(SEED + 3)
    b = W.OceanBatch(n, 1, 0)
    assert _abi.lib().ocean_select_streams(b._h, 10, None) == _abi.OCEAN_E_NOT_READY        # before ocean_prepare
    b.prepare(SEED + 3)
    # with caller-bound output the calibration frames would land in the caller's memory: refused, and the memory stays untouched
    import torch
    mine = torch.full((2, n, n, 4), -7.0, dtype=torch.float32, device="cuda:0")
    b.bind_output(mine[0].data_ptr(), mine[1].data_ptr())
    assert _abi.lib().ocean_select_streams(b._h, 10, None) == _abi.OCEAN_E_UNSUPPORTED
    b.synchronize()
    assert bool((mine == -7.0).all())
    b.bind_output(None, None)
    a0 = b.compute_waves(0.5)
    us = b.select_streams(30)
    assert len(us) == 4 and all(u > 0 for u in us) and us == sorted(us)
    assert max(us) < 3 * min(us)                                  # four queues of one device: the same order of magnitude
    with pytest.raises(_abi.OceanError):                          # the calibration frame is not a result
        b.read_maps()
    assert np.array_equal(a0, ref.compute_waves(0.5))
    for t in (0.5, 1.75):
        a = b.compute_waves(t)
        d, q = b.read_maps()
        ar = ref.compute_waves(t)
        dr, qr = ref.read_maps()
        assert np.array_equal(a, ar) and np.array_equal(d, dr) and np.array_equal(q, qr)
    b.set_pipeline_depth(3)
    b.set_frame_tracking(True)
    for j in range(7):
        b.compute_waves_async(0.3 * j)
    a = b.wait_frame()
    b.synchronize()
    d, q = b.read_maps()
    ar = ref.compute_waves(0.3 * 6)
    dr, qr = ref.read_maps()
    assert np.array_equal(a, ar) and np.array_equal(d, dr) and np.array_equal(q, qr)
    us2 = b.select_streams(30)                                    # again, at depth 3: still serial calibration frames, same contract
    assert us2 == sorted(us2)
    b.compute_waves_async(2.5); b.synchronize()
    ref.compute_waves(2.5)
    assert np.array_equal(b.read_maps()[0], ref.read_maps()[0])
    b.close(); ref.close()


def test_round5_switches_change_launches_and_no_bit():
    """The switches round 5 added for what the library otherwise decides (include/ocean.h): ocean_set_start_ramp(0)
    takes the staggered start off a 2048^2 frame's launches, ocean_set_merged_xpass(0) gives pipelined small tiles their three launches back,
    ocean_set_external_readers(0 / 1) moves the synchronous call between the completion-record poll and the stream synchronisation after
    ocean_device_maps -- and none of them changes a bit of a frame; ocean_build_id() is the hash of the sources beside the library."""
    import watersurfacerendering_amd as W
    from watersurfacerendering_amd import _abi as A
    assert A.lib().ocean_build_id().decode() == A.source_build_id() == A.library_build_id()
    # staggered start
    n = 2048
    out = []
    for on in (True, False):
        b = W.OceanBatch(n, 1, 0)
        b.set_start_ramp(on)
        b.prepare(SEED + 11)
        amp = b.compute_waves(0.9)
        flags = [li["flags"] for li in b.last_launch()]
        assert all(bool(f & A.OCEAN_LAUNCH_STAGGERED_START) == on for f in flags), (on, flags)
        assert bool(flags[0] & A.OCEAN_LAUNCH_WT_INTER)                   # (a serial 2048^2 frame: write-through intermediates either way)
        out.append((amp, *b.read_maps()))
        b.close()
    assert all(np.array_equal(x, y) for x, y in zip(out[0], out[1]))
    # merged x pass: pipelined 512^2 frames, on / off
    n = 512
    out = []
    for on in (True, False):
        b = W.OceanBatch(n, 1, 0)
        b.set_merged_xpass(on); b.set_pipeline_depth(3)
        b.prepare(SEED + 12)
        for j in range(5):
            b.compute_waves_async(0.2 * j)
        b.synchronize()
        assert all(bool(li["flags"] & A.OCEAN_LAUNCH_MERGED_X) == on for li in b.last_launch()[1:])
        out.append(b.read_maps())
        b.close()
    assert all(np.array_equal(x, y) for x, y in zip(out[0], out[1]))
    # external readers: the flag ocean_device_maps sets can be taken back (and set by hand); frames and amplitudes are the same in both wait modes
    b = W.OceanBatch(256, 1, 0)
    b.prepare(SEED + 13)
    a0 = b.compute_waves(0.3)
    b.device_maps()                                      # -> the synchronous call synchronises the stream from here on
    a1 = b.compute_waves(0.3)
    b.set_external_readers(False)                        # every consumer is on ocean_stream(): back to the poll
    a2 = b.compute_waves(0.3)
    b.set_external_readers(True)
    a3 = b.compute_waves(0.3)
    assert np.array_equal(a0, a1) and np.array_equal(a0, a2) and np.array_equal(a0, a3)
    b.close()


@pytest.mark.parametrize("n,tiles,depth", [(64, 1, 1), (128, 1, 1), (512, 1, 1), (1024, 1, 1), (2048, 1, 1), (256, 3, 1), (512, 1, 3), (1024, 2, 2)])
def test_compute_waves_read_is_compute_waves_plus_read_maps(n, tiles, depth):
    """ocean_compute_waves_read (round 6): synthesis + both maps of every tile in host memory as ONE blocking call -- the reference's call
    shape, WaterSurfaceMesh.cpp:145-154 + 701-755 -- with the normal map's copy started behind the frame's second launch.  Bit for bit what
    ocean_compute_waves + ocean_read_maps deliver: small tiles (whose x axis is one launch: no such point in the frame, both copies follow
    it), the reference's default 512^2, the headline 2048^2, batches, pipelined contexts; pageable and page-locked destinations; repeated
    calls into the same arrays (a copy still in flight would show)."""
    import watersurfacerendering_amd as W
    from watersurfacerendering_amd import _abi as A
    b = W.OceanBatch(n, tiles, 0)
    b.set_pipeline_depth(depth)
    b.prepare(SEED + 21)
    ref = []
    for t in (0.0, 1.25, 7.5):
        a = b.compute_waves(t)
        ref.append((a.copy(), *[x.copy() for x in b.read_maps()]))
    # pageable destinations, fresh arrays
    for t, (a0, d0, q0) in zip((0.0, 1.25, 7.5), ref):
        a, d, q = b.compute_waves_read(t)
        assert np.array_equal(a, a0) and np.array_equal(d, d0) and np.array_equal(q, q0), t
        assert b.heights(tiles - 1)[0] == a0[tiles - 1]
    # page-locked destinations, the same two arrays every call (the adaptor's use)
    d = pinned_array((tiles, n, n, 4), np.float32, tag="d"); q = pinned_array((tiles, n, n, 4), np.float32, tag="q")
    L = A.lib()
    if True:
        for rep in range(3):
            for t, (a0, d0, q0) in zip((0.0, 1.25, 7.5), ref):
                a, _, _ = b.compute_waves_read(t, d, q)
                assert np.array_equal(a, a0) and np.array_equal(d, d0) and np.array_equal(q, q0), (rep, t)
        # mixed with the other calls of the frame path
        b.compute_waves_async(3.0); b.compute_waves_async(4.0)
        a, _, _ = b.compute_waves_read(1.25, d, q)
        assert np.array_equal(a, ref[1][0]) and np.array_equal(d, ref[1][1]) and np.array_equal(q, ref[1][2])
        dd, qq = b.read_maps()
        assert np.array_equal(dd, ref[1][1]) and np.array_equal(qq, ref[1][2])
    assert L.ocean_compute_waves_read(b._h, 0.0, None, None, q.ctypes.data) == A.OCEAN_E_INVALID
    assert b.fault_recoveries == 0
    b.close()


def test_placement_search_changes_no_bit_and_reports():
    """ocean_prepare's placement search (round 6; include/ocean_dev.h): from 2048^2 up Prepare times a few candidate allocations of the spectrum +
    intermediates on serial frames and keeps the fastest -- the same 2048^2 z pass runs 19.8 ... 28.6 us depending on where those buffers landed
    (profiles/r06_slow_window.txt).  Frames do not depend on it bit for bit; the report says what was done; below 2048^2 nothing is searched
    unless asked for; the calibration frames leave nothing to read out; a repeated Prepare on the same buffers does not search again."""
    import watersurfacerendering_amd as W
    from watersurfacerendering_amd import _abi as A
    for n, tiles in ((1024, 1), (2048, 1), (1024, 3)):
        frames = []
        for trials in (1, 0, 9):
            b = W.OceanBatch(n, tiles, 0)
            b.set_placement_search(trials)
            b.prepare(SEED + 31)
            tried, chosen, worst = b.placement_report()
            if trials == 1 or (trials == 0 and n < 2048):
                assert (tried, chosen, worst) == (0, 0.0, 0.0)
            else:
                assert tried == (6 if trials == 0 else 9) and 0.0 < chosen <= worst < 10.0 * chosen, (n, tiles, trials, tried, chosen, worst)
            with pytest.raises(W.OceanError) as e:            # the maps hold a calibration frame: nothing to read out before the first frame
                b.read_maps()
            assert e.value.code == A.OCEAN_E_NOT_READY
            a = b.compute_waves(1.75)
            d, q = b.read_maps()
            h0, om = b.read_spectrum(tiles - 1)
            b.set_pipeline_depth(2)
            for j in range(4):
                b.compute_waves_async(0.1 * j)
            b.compute_waves_async(1.75); b.synchronize()
            d2, q2 = b.read_maps()
            assert np.array_equal(d, d2) and np.array_equal(q, q2)
            frames.append((a, d, q, h0, om))
            if trials == 9:
                # a second Prepare on the same buffers (a parameter change in the reference's GUI) does not search again -- a placement keeps
                # its speed for as long as it lives -- keeps the report and delivers the same ocean; another candidate count or a resize does
                import time
                t0 = time.perf_counter(); b.prepare(SEED + 31); again_ms = (time.perf_counter() - t0) * 1e3
                assert b.placement_report() == (tried, chosen, worst)
                a3 = b.compute_waves(1.75); d3, q3 = b.read_maps()
                assert np.array_equal(a3, a) and np.array_equal(d3, d) and np.array_equal(q3, q)
                b.set_placement_search(4)
                t0 = time.perf_counter(); b.prepare(SEED + 31); search_ms = (time.perf_counter() - t0) * 1e3
                assert b.placement_report()[0] == 4 and search_ms > again_ms + 5.0, (again_ms, search_ms)
                b.set_tile_size(n // 2); b.set_tile_size(n); b.prepare(SEED + 31)
                assert b.placement_report()[0] == 4
                a4 = b.compute_waves(1.75); d4, q4 = b.read_maps()
                assert np.array_equal(a4, a) and np.array_equal(d4, d) and np.array_equal(q4, q)
            b.close()
        for other in frames[1:]:
            assert all(np.array_equal(x, y) for x, y in zip(frames[0], other)), (n, tiles)
    b = W.OceanBatch(512, 1, 0); b.prepare(SEED)
    assert b.placement_report()[0] == 0                          # the reference's default size: Prepare stays as cheap as it was
    b.set_placement_search(3); b.prepare(SEED)
    assert b.placement_report()[0] == 3
    b.close()


@isolated
def test_host_register_round_trip():
    """ocean_host_register / ocean_host_unregister on a small page-aligned range: both succeed once, the second unregistration is refused by
    the runtime (reported, not fatal), and the library's own list of ranges forgets the entry (no stale device address is handed out).  The one
    place in the suite that unregisters -- in a process of its own: see conftest.pinned_array."""
    import watersurfacerendering_amd as W
    from watersurfacerendering_amd import _abi as A
    L = A.lib()
    raw = np.zeros(65536 + 8192, dtype=np.uint8)
    o = (-raw.ctypes.data) % 4096
    a = raw[o:o + 65536]
    assert L.ocean_host_register(a.ctypes.data, a.nbytes) == A.OCEAN_OK
    assert L.ocean_host_unregister(a.ctypes.data) == A.OCEAN_OK
    assert L.ocean_host_unregister(a.ctypes.data) == A.OCEAN_E_HIP
    assert L.ocean_host_unregister(None) == A.OCEAN_E_INVALID
    # a context still works with pageable destinations of that very range afterwards
    b = W.OceanBatch(64, 1, 0); b.prepare(3)
    d = a[:64 * 64 * 16].view(np.float32).reshape(1, 64, 64, 4); q = np.empty_like(d)
    amp, _, _ = b.compute_waves_read(0.5, d, q)
    d2, q2 = b.read_maps()
    assert np.array_equal(d, d2) and np.array_equal(q, q2)
    b.close()


@pytest.mark.parametrize("n", [256, 1024])
def test_compute_waves_read_in_every_mode_and_binding(n):
    """ocean_compute_waves_read in the configurations the size sweep above leaves out: every mode (the x passes' host stores are written per
    role: HEIGHT1 emits zero slopes / displacements, JACOBIAN a fourth channel), half2 intermediates, the 16-bit spectrum, per-tile time
    offsets, a caller-owned stream and caller-bound output maps -- page-locked destinations (direct stores at 256^2, engine copies at
    1024^2) and pageable ones; always the bits ocean_compute_waves + ocean_read_maps deliver."""
    import torch
    import watersurfacerendering_amd as W
    from watersurfacerendering_amd import _abi as A
    tiles = 2
    d = pinned_array((tiles, n, n, 4), np.float32, tag="d"); q = pinned_array((tiles, n, n, 4), np.float32, tag="q")

    def check(b, what):
        for t in (0.0, 2.5):
            a0 = b.compute_waves(t).copy(); d0, q0 = b.read_maps()
            a, _, _ = b.compute_waves_read(t, d, q)
            assert np.array_equal(a, a0) and np.array_equal(d, d0) and np.array_equal(q, q0), (what, t, "page-locked")
            a, dp, qp = b.compute_waves_read(t)
            assert np.array_equal(a, a0) and np.array_equal(dp, d0) and np.array_equal(qp, q0), (what, t, "pageable")

    b = W.OceanBatch(n, tiles, 0)
    for mode in (A.OCEAN_MODE_FULL7, A.OCEAN_MODE_CHOPPY5, A.OCEAN_MODE_HEIGHT1, A.OCEAN_MODE_JACOBIAN):
        b.set_mode(mode); b.prepare(SEED + 31)
        check(b, f"mode {mode}")
    b.set_mode(A.OCEAN_MODE_FULL7)
    b.set_intermediate_precision(16); b.prepare(SEED + 31); check(b, "half2 intermediates")
    b.set_intermediate_precision(32); b.set_spectrum_precision(16); b.prepare(SEED + 31); check(b, "16-bit spectrum")
    b.set_spectrum_precision(32); b.set_time_offsets([0.0, 3.5]); b.prepare(SEED + 31); check(b, "time offsets")
    b.set_time_offsets(None)
    s = torch.cuda.Stream()
    b.set_stream(s.cuda_stream); b.prepare(SEED + 31); check(b, "caller's stream")
    b.set_stream(None)
    maps = torch.zeros((2, tiles, n, n, 4), dtype=torch.float32, device="cuda:0")
    b.bind_output(maps[0].data_ptr(), maps[1].data_ptr()); b.prepare(SEED + 31); check(b, "bound output")
    a, _, _ = b.compute_waves_read(1.0, d, q)
    torch.cuda.synchronize()
    assert np.array_equal(maps[0].cpu().numpy(), d) and np.array_equal(maps[1].cpu().numpy(), q)       # the device copy stays complete
    b.bind_output(None, None)
    assert b.fault_recoveries == 0
    b.close()


def test_compute_waves_read_with_memory_page_locked_elsewhere():
    """Destinations page-locked by other means than ocean_host_register: a pinned torch tensor (hipHostMalloc) takes the direct-store path through
    the runtime's lookup.  Same bits as call + read-out."""
    import torch
    import watersurfacerendering_amd as W
    n, tiles = 256, 2
    b = W.OceanBatch(n, tiles, 0); b.prepare(SEED + 41)
    a0 = b.compute_waves(0.75).copy(); d0, q0 = b.read_maps()
    td = torch.empty((tiles, n, n, 4), dtype=torch.float32, pin_memory=True); tq = torch.empty_like(td, pin_memory=True)
    for t in (0.75, 0.75):
        a, _, _ = b.compute_waves_read(t, td.numpy(), tq.numpy())
        assert np.array_equal(a, a0) and np.array_equal(td.numpy(), d0) and np.array_equal(tq.numpy(), q0)
    assert b.fault_recoveries == 0
    b.close()


@isolated
def test_compute_waves_read_refuses_a_partly_registered_destination():
    """An array of which only the first half is page-locked must NOT take the direct-store path -- the x passes would store past the mapping --:
    the call goes to the copy engines, whose copy the runtime refuses (a destination that straddles registered and unregistered memory);
    reported, nothing faults, the context goes on.  In a process of its own: it leaves a half-registered allocation behind."""
    import ctypes as C
    import watersurfacerendering_amd as W
    n, tiles = 256, 2
    b = W.OceanBatch(n, tiles, 0); b.prepare(SEED + 41)
    a0 = b.compute_waves(0.75).copy(); d0, q0 = b.read_maps()
    hip = C.CDLL("libamdhip64.so")
    raw = np.zeros(tiles * n * n * 16 + 8192, dtype=np.uint8)
    o = (-raw.ctypes.data) % 4096
    d = raw[o:o + tiles * n * n * 16].view(np.float32).reshape(tiles, n, n, 4)
    assert hip.hipHostRegister(C.c_void_p(d.ctypes.data), C.c_size_t(d.nbytes // 2), 0) == 0       # the first half only; stays registered
    q = np.empty_like(d0)
    for dst in ((d, q), (q, d)):
        with pytest.raises(W.OceanError) as e:
            b.compute_waves_read(0.75, *dst)
        assert e.value.code == W._abi.OCEAN_E_HIP
    b.synchronize()
    a, dd, qq = b.compute_waves_read(0.75)
    assert np.array_equal(a, a0) and np.array_equal(dd, d0) and np.array_equal(qq, q0)
    assert b.fault_recoveries == 0
    b.close()
