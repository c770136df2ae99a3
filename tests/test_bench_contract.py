"""bench.py's CPU-checkable contract: flag defaults, byte accounting, and the CPU baseline leg on a tiny size."""
import importlib
import sys

import pytest


@pytest.fixture()
def bench(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    return importlib.import_module("bench")


def test_defaults_are_one_gpu_and_bounded(bench):
    a = bench.parse()
    assert a.gpus == 1 and a.steps > 0 and a.warmup >= 0
    assert a.size == 2048 and a.tiles == 1            # BASELINE.json: the 2048x2048 single-tile configuration
    assert 1 <= a.depth <= 8


def test_byte_model_matches_survey_8d(bench):
    # 12 (h0 + omega) + 16 * 3.5 (intermediates out and in) + 32 (maps) + 8 (height normalisation)
    assert bench.FRAME_BYTES_SURVEY == 12 + 16 * 3.5 + 32 + 8 == 108
    assert sum(bench.KERNEL_BYTES_SURVEY.values()) == 108
    assert sum(bench.KERNEL_BYTES_ACTUAL.values()) == 74
    assert set(bench.KERNEL_BYTES_SURVEY) == set(bench.KERNEL_BYTES_ACTUAL) == {"k_zpass", "k_xpass_b", "k_xpass_disp"}
    assert bench.HBM_PEAK_GBPS == 8000.0


def test_cpu_baseline_leg_runs_and_reports_its_shape(bench):
    r = bench.cpu_baseline(64, 0.5)
    assert r["kind"] == "port" and r["unit"] == "frames/s" and r["value"] > 0
    assert r["cores"] >= 1 and "sample" in r and "FFTW not available" in r["sample"]
    assert r["host"]["nproc"] >= 1


def test_committed_traffic_file_matches_the_kernels(bench):
    """profiles/traffic.json (PMC-derived HBM bytes per launch) must name the kernels bench.py accounts for."""
    import json
    import os
    t = json.load(open(os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "profiles", "traffic.json")))
    assert set(t) == {k + "@2048" for k in bench.KERNEL_BYTES_SURVEY}
    n2 = 2048 * 2048
    for k, v in t.items():
        own = bench.KERNEL_BYTES_ACTUAL[k.split("@")[0]] * n2
        assert 0.9 * own <= v["hbm_bytes_per_launch"] <= 1.15 * own, (k, v["hbm_bytes_per_launch"], own)   # no wasted re-reads
        assert os.path.exists(os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), v["source"].split(" ")[0]))
