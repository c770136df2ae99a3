import functools
import os
import subprocess
import sys
import warnings

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

    For the tests that share device memory between processes (dma-buf export / import) or drive the device out of memory: on this
    pool the HIP runtime now and then abort()s -- silently, inside a later, unrelated runtime call (a read-out, a prepare) of the process
    that exported, imported or exhausted memory: 3 of ~20 sessions in round 4, never under a debugger hook, never when such a test ran
    alone -- and an abort takes the whole pytest session with it.  In its own process the test checks exactly what it checked before;
    a child that DIES FROM A SIGNAL inside the runtime (not one that fails an assertion) is run once more and reported with a warning,
    a second death fails the test.  The session's other tests never touch inter-process memory and are not exposed.
    (tools/abort_hunt.sh runs the sessions un-isolated again, with the interpreter's fault handler and the runtime's error log: 14 full sessions and
    15 of tests/test_api_state_gpu.py alone on one box at the end of round 4, no abort -- it depends on the box, and stays unexplained.)"""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if os.environ.get("OCEAN_TEST_CHILD") == "1":
            return fn(*args, **kwargs)
        node = os.path.relpath(fn.__code__.co_filename, ROOT) + "::" + fn.__name__
        cmd = [sys.executable, "-m", "pytest", node, "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"]
        env = dict(os.environ, OCEAN_TEST_CHILD="1")
        for attempt in (1, 2):
            r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
            if r.returncode == 0:
                assert " passed" in r.stdout, r.stdout[-2000:]          # (really ran: not deselected, not skipped)
                return
            died = r.returncode < 0 or r.returncode in (134, 139)
            if died and attempt == 1:
                warnings.warn(f"{node}: the child process died inside the runtime (status {r.returncode}); running it once more")
                continue
            pytest.fail(f"{node} in its own process: status {r.returncode}\n{r.stdout[-3000:]}\n{r.stderr[-2000:]}", pytrace=False)
    return wrapper
