"""The shape of bench.py's stdout line, in ONE place.

Round 5 flattened the line's `gather` object (`compute_only.tiles_per_s` -> `compute_only_tiles_per_s`) and updated the CPU
contract test but not the multi-GPU one, which only runs on a node nobody had: the first 2-GPU run would have died in the harness
with a KeyError.  Both tests now go through the functions below -- tests/test_bench_contract.py feeds them a line that
`bench.build_line` builds from a fully populated world-8 run (runs everywhere, every round), tests/test_zz_multi_gpu.py feeds them
the line a real N-GPU run printed -- so a rename fails on the CPU box first.
"""
import json

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline")
CONFIG_KEYS = ("workload", "tile_size", "tiles_per_rank", "pipeline_depth", "parallelism")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launch_us", "rocprof_launch_us", "frame_frac",
                 "serial_frame_frac", "kernels")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample")
# the compact `gather` object: what bench.compact_gather writes from measure_gather's dictionary
GATHER_SCALARS = ("ranks", "rccl_ranks_seen", "tile_size", "tiles_per_rank", "bytes_into_root_per_step", "root_copy_matches_local_maps")
GATHER_REGIMES = ("compute_only", "compute_plus_gather_serial", "compute_gather_overlapped", "compute_gather_overlapped_half_maps")
GATHER_RATE_KEYS = tuple(k + "_tiles_per_s" for k in GATHER_REGIMES)


def last_json_line(stdout: str) -> dict:
    """The one JSON line of a bench.py run (everything else goes to stderr)."""
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line on stdout, found {len(lines)}"
    return json.loads(lines[0], parse_constant=lambda c: (_ for _ in ()).throw(AssertionError(f"non-strict JSON constant {c}")))


def check_contract_line(out: dict, world: int) -> None:
    """Keys and invariants of every line, at any N."""
    for key in CONTRACT_KEYS:
        assert key in out, f"line lacks {key!r}"
    assert out["n_gpus"] == world and out["value"] > 0 and out["ms_per_step"] > 0
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["dtype"] == "f32" and out["data"] == "synthetic" and out["unit"] == "frames/s"
    for key in CONFIG_KEYS:
        assert key in out["config"], f"config lacks {key!r}"
    assert "model" not in out["config"]
    r = out["roofline"]
    for key in ROOFLINE_KEYS:
        assert key in r, f"roofline lacks {key!r}"
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-4 * r["frac"] and 0 < r["frac"] < 1.0
    assert r["kernel"] in r["kernels"]
    c = out["cpu_baseline"]
    if c is not None:                                  # (--no-cpu-baseline runs carry null)
        for key in CPU_KEYS:
            assert key in c, f"cpu_baseline lacks {key!r}"
        assert c["value"] > 0 and c["kind"] in ("port", "reference") and c["cores"] >= 1


def check_gather(g: dict, world: int, tiles_per_rank: int = 8, tile_size: int = 1024) -> None:
    """The compact `gather` object of a run whose exchange step completed on `world` ranks."""
    assert isinstance(g, dict) and "error" not in g, g
    for key in GATHER_SCALARS + GATHER_RATE_KEYS:
        assert key in g, f"gather lacks {key!r}"
    assert g["ranks"] == world and g["rccl_ranks_seen"] == world
    assert g["root_copy_matches_local_maps"] is True
    assert g["tile_size"] == tile_size and g["tiles_per_rank"] == tiles_per_rank
    assert g["bytes_into_root_per_step"] == (world - 1) * tiles_per_rank * tile_size * tile_size * 32
    for key in GATHER_RATE_KEYS:
        assert isinstance(g[key], (int, float)) and g[key] > 0, key
    if world > 1:
        # every gathered regime moves (world-1) x 256 MiB into the root per step over xGMI: it cannot beat compute alone
        assert g["compute_only_tiles_per_s"] > g["compute_plus_gather_serial_tiles_per_s"]


def check_multi_gpu_line(out: dict, world: int) -> None:
    """What tests/test_zz_multi_gpu.py asserts of a real N > 1 run, and tests/test_bench_contract.py of a synthetic one."""
    check_contract_line(out, world)
    check_gather(out["gather"], world)
    assert out["config"]["parallelism"].endswith(f"x{world}, no data-path collective")
    if (out["config"]["tile_size"], out["config"]["tiles_per_rank"], world) == (1024, 8, 8):
        assert "BASELINE config 5" in out["config"]["workload"]
