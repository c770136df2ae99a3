import functools
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def isolated(fn):
    """Run a GPU test in a pytest process of its own.

    For the tests that share device memory between processes (dma-buf export / import) or drive the device out of memory: they change
    process-wide state of the HIP runtime (imported memory objects, a failed 8 TB allocation) that the session's other tests have no
    business inheriting, and a crash of the runtime in one of them must be attributed to THAT test.  In its own process the test checks
    exactly what it checked before.  A child that dies from a signal FAILS the test at once, with the interpreter's fault-handler
    traceback and the child's output attached -- it is never run again (round 4 re-ran it once and only warned: an intermittent silent
    abort() of the runtime, 3 of ~20 sessions on some boxes, was thereby retried away; INTEGRATION.md, known issues).
    OCEAN_TEST_SIGNAL_XFAIL=1 reports such a death as a non-strict xfail instead (for a box known to show the runtime issue); nothing
    sets it by default.  tools/abort_hunt.sh runs the sessions un-isolated for a hunt (round 5: 50 of 50 clean)."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if os.environ.get("OCEAN_TEST_CHILD") == "1":
            return fn(*args, **kwargs)
        node = os.path.relpath(fn.__code__.co_filename, ROOT) + "::" + fn.__name__
        cmd = [sys.executable, "-X", "faulthandler", "-m", "pytest", node, "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"]
        env = dict(os.environ, OCEAN_TEST_CHILD="1")
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        if r.returncode == 0:
            assert " passed" in r.stdout, r.stdout[-2000:]          # (really ran: not deselected, not skipped)
            return
        died = r.returncode < 0 or r.returncode in (134, 139)
        report = f"{node} in its own process: status {r.returncode}{' (died from a signal)' if died else ''}\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
        if died and os.environ.get("OCEAN_TEST_SIGNAL_XFAIL") == "1":
            pytest.xfail(report)
        pytest.fail(report, pytrace=False)
    return wrapper


_PINNED_POOL = {}


def pinned_array(shape, dtype, fill=0, tag=""):
    """A page-locked numpy array from a process-wide pool: registered ONCE through ocean_host_register and kept until the process ends -- the
    adaptor's pattern (include/WSTessendorf.hpp registers its vectors once per size).  The GPU suite does not register and unregister host
    ranges test by test: on this ROCm stack a process that churns registrations while it also copies into pageable memory now and then dies
    of a page fault inside the runtime's on-the-fly pinning (profiles/r06_hostreg_churn_fault.txt; tools/soak_api.py reproduces it with
    SOAK_SKIP=churn, 10-30 % of 6000-operation runs; with persistent registrations 0 of 32)."""
    import numpy as np
    import watersurfacerendering_amd as W
    key = (tuple(shape) if hasattr(shape, "__len__") else (int(shape),), np.dtype(dtype).str, tag)
    if key not in _PINNED_POOL:
        a = np.empty(key[0], dtype=dtype)
        W.host_register(a)
        _PINNED_POOL[key] = a
    a = _PINNED_POOL[key]
    a[...] = fill
    return a
