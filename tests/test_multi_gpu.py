"""Multi-GPU tests of the tile-sharded batch mode: they need at least two MI355X in the node and are
skipped otherwise (a gpurun box has one).  One process per GPU, the library's own RCCL communicator
(ocean_comm_init / ocean_gather_maps), every rank's tiles compared on the root bit for bit."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpus() -> int:
    import torch
    return torch.cuda.device_count()


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "OCEAN_BENCH_BACKEND")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_rccl_gather_of_sharded_tiles(world):
    if _gpus() < world:
        pytest.skip(f"needs {world} GPUs, node has {_gpus()}")
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tests", "workers", "gather_worker.py")], capture_output=True, text=True, env=_env(), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "GATHER_OK" in r.stdout, r.stdout[-2000:]


def test_bench_two_gpus_over_rccl():
    if _gpus() < 2:
        pytest.skip("needs 2 GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "200", "--warmup", "50", "--prewarm", "100",
                        "--no-extra", "--no-cpu-baseline"], capture_output=True, text=True, env=_env(), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0
    g = out["gather"]
    assert g["ranks"] == 2 and g["root_copy_matches_local_maps"] is True
    assert g["compute_only"]["tiles_per_s"] > g["compute_plus_gather_serial"]["tiles_per_s"] > 0
