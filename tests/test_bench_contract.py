"""bench.py's CPU-checkable contract: flag defaults, byte accounting, the self-launcher and the CPU baseline leg on a tiny size."""
import importlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    return importlib.import_module("bench")


def test_defaults_are_one_gpu_and_bounded(bench):
    a = bench.parse([])
    assert a.gpus == 1 and a.steps > 0 and a.warmup >= 0
    assert a.size == 2048 and a.tiles == 1            # BASELINE.json: the 2048x2048 single-tile configuration
    assert 1 <= a.depth <= 8


def test_byte_accounting(bench):
    # this pipeline's own bytes are what roofline.achieved uses ...
    assert sum(bench.KERNEL_BYTES_ACTUAL.values()) == bench.FRAME_BYTES_ACTUAL == 73
    # ... SURVEY.md 8d's model is kept beside it: 12 (h0 + omega) + 16 * 3.5 (intermediates out and in) + 32 (maps) + 8
    assert bench.FRAME_BYTES_SURVEY == 12 + 16 * 3.5 + 32 + 8 == 108
    assert sum(bench.KERNEL_BYTES_SURVEY.values()) == 108
    assert set(bench.KERNEL_BYTES_SURVEY) == set(bench.KERNEL_BYTES_ACTUAL) == {"k_zpass", "k_xpass_b", "k_xpass_disp"}
    assert bench.HBM_PEAK_GBPS == 8000.0


def test_roofline_object_uses_own_bytes_and_never_exceeds_peak_on_them(bench):
    names = ["k_zpass", "k_xpass_b", "k_xpass_disp"]
    n2 = 2048 * 2048
    r = bench.roofline_object(2048, 1, names, [0.027, 0.029, 0.0205], [0.06, 0.05, 0.04], 3, 0.054, 77.0, 73.0)
    assert r["kernel"] == "k_xpass_b" and r["algorithmic_bytes_per_launch"] == 28 * n2
    assert r["achieved"] == pytest.approx(28 * n2 / 29e-6 * 1e-9) and r["frac"] == pytest.approx(r["achieved"] / 8000.0)
    assert r["frame_frac"] == pytest.approx(73 * n2 / 54e-6 * 1e-9 / 8000.0) and r["frame_frac"] < 1.0
    assert "MODEL" in r["survey_model"]["what"] and "frac" not in r["survey_model"]       # the 108 figure is never a fraction of peak
    assert set(r["kernels"]) == set(names)


def test_multi_gpu_line_carries_config5_workload_cpu_baseline_and_roofline(bench):
    """What rank 0 prints at N > 1: the documented config-5 invocation names the workload as BASELINE does,
    and the line keeps `cpu_baseline` and `roofline` (round 2 dropped both at N > 1)."""
    a = bench.parse(["--gpus", "8", "--size", "1024", "--tiles", "8", "--depth", "2", "--steps", "100", "--warmup", "20"])
    names = ["k_zpass", "k_xpass_b", "k_xpass_disp"]
    roof = bench.roofline_object(1024, 8, names, [0.045, 0.04, 0.03], [0.06, 0.05, 0.04], 2, 0.11, 120.0, 73.0,
                                 dict(zip(names, (23, 28, 22))))
    cpu = {"value": 20.0, "unit": "frames/s", "cores": 16, "kind": "port", "sample": "synthetic"}
    gather = {"ranks": 8, "rccl_ranks_seen": 8, "compute_only": {"tiles_per_s": 5e5}}
    line = bench.build_line(a, 8, 1024, 8, 5.6e5, 0.11, roof, gather, {}, cpu, dict(cpu, value=80.0))
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["vs_baseline"] is None and line["dtype"] == "f32"
    assert line["config"]["workload"].startswith("64 x 1024x1024 tiles, 8 per GPU") and "BASELINE config 5" in line["config"]["workload"]
    assert line["config"]["tiles_per_rank"] == 8 and line["config"]["pipeline_depth"] == 2
    assert line["cpu_baseline"]["value"] == 20.0 and line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline_strong"]["value"] == 80.0
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and r["kernel"] in names
    assert r["algorithmic_bytes_per_launch"] == r["kernel_bytes_per_texel"][r["kernel"]] * 1024 * 1024 * 8
    assert line["gather"]["rccl_ranks_seen"] == 8
    assert line["warmup"] == 20 and line["warmup_frames_effective"] == 20 + a.prewarm
    json.dumps(line)
    # a single-GPU single-tile line still names the headline tile
    one = bench.build_line(bench.parse([]), 1, 2048, 1, 2e4, 0.05, roof, None, {}, cpu, None)
    assert one["config"]["workload"].startswith("2048x2048 tile")


def test_stale_profile_summaries_are_not_quoted(bench, monkeypatch):
    """traffic / rocprof figures of profiles/*.json enter the line only while the kernel sources hash to what they were measured with."""
    names = ["k_zpass", "k_xpass_b", "k_xpass_disp"]
    r = bench.roofline_object(2048, 1, names, [0.025, 0.022, 0.0175], None, 1, 0.066, 66.0, 73.0)
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    current = t.get("_kernel_source_sha16") == bench.kernel_source_sha16()
    assert r["committed_profiles"]["current"] == current
    assert (r["traffic"] is not None) == current
    monkeypatch.setattr(bench, "kernel_source_sha16", lambda: "0" * 16)
    r = bench.roofline_object(2048, 1, names, [0.025, 0.022, 0.0175], None, 1, 0.066, 66.0, 73.0)
    assert r["traffic"] is None and r["rocprof_launch_us"] is None and all("traffic_bytes_per_launch" not in k for k in r["kernels"].values())


def test_cpu_baseline_leg_runs_and_reports_its_shape(bench):
    ref, strong = bench.cpu_baseline(64, 0.3)
    assert ref["kind"] == "port" and ref["unit"] == "frames/s" and ref["value"] > 0
    assert ref["cores"] >= 1 and "sample" in ref and ("FFTW not available" in ref["sample"] or ref["fft"] == "fftw3f")
    assert ref["host"]["nproc"] >= 1
    # (the strongest candidate of THIS run: the oracle's own FFT work-shared by a team, or scipy's pocketfft with stage D shared out -- which one
    #  wins depends on the host and on its load)
    assert strong["value"] > 0 and ("work-shared" in strong["sample"] or "shared out" in strong["sample"])
    # `cores` is the USABLE CPU count (min of the affinity mask and the cgroup quota) -- the team the sample really ran
    # with -- and the host's logical CPU count stands beside it (round 3 printed cores = 256 next to a quota of 16)
    usable, quota, aff, nproc = bench.usable_cpus()
    assert 1 <= usable <= aff <= nproc and (quota is None or usable <= int(quota + 0.999999))
    assert ref["cores"] == min(usable, ref["host"]["omp_max_threads"]) and ref["nproc"] == nproc == ref["host"]["nproc"]
    assert ref["host"]["usable_cpus"] == usable and f"{ref['cores']} threads" in ref["sample"]


def test_timed_regions_are_repeated_and_reported_as_median_with_spread(bench):
    """SURVEY.md 8d / VERDICT r03 #4: the timed region of --steps frames is repeated K >= 7 times inside one invocation; ms_per_step is the
    median region, p10 / p90 stand beside it.  (The GPU side is exercised by tests/test_parity_bench_regimes.py; here: the defaults and
    that the line keeps the contract's keys with the statistics added.)"""
    a = bench.parse([])
    assert a.regions >= 7
    assert bench.parse(["--regions", "11"]).regions == 11
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ("ms_per_step_p10", "ms_per_step_p90", "ms_per_step_median", "steps_per_region", "p10_ms_per_step", "p90_ms_per_step"):
        assert key in src
    names = ["k_zpass", "k_xpass_b", "k_xpass_disp"]
    roof = bench.roofline_object(2048, 1, names, [0.025, 0.022, 0.0175], None, 1, 0.066, 66.0, 73.0)
    line = bench.build_line(a, 1, 2048, 1, 2e4, 0.05, roof, None, {}, None, None)
    assert line["steps"] == a.steps and line["warmup"] == a.warmup and line["ms_per_step"] == 0.05


def test_config4_check_compares_saved_gpu_rows_with_the_float64_oracle(bench, tmp_path):
    """The CPU-baseline child's second job (VERDICT r03 #6a): frames the GPU side saved are compared with the float64 oracle per variant.
    Here the 'GPU frames' are the oracle's own float32-FFT frame (error at the fp32 floor) and a perturbed copy (error visible)."""
    import numpy as np
    from oracle import oracle as O
    n, t = 64, 1.0
    xi = O.gauss_xi_numpy(bench.SEED, n)
    o = O.Oracle(n)
    o.prepare(xi=xi)
    amp, d, q = o.compute_waves(t, fft=O.FFT_F32)
    rows = np.arange(0, n, 8)
    maps = np.concatenate([d[rows], q[rows]], axis=-1)
    bad = maps.copy(); bad[..., 0] += 1e-2 * np.abs(d[..., 0]).max()
    path = tmp_path / "check.npz"
    np.savez(path, n=n, t=t, rows=rows, names=np.array(["fp32", "broken"]), xi=xi, maps_fp32=maps, amp_fp32=amp, maps_broken=bad, amp_broken=amp)
    r = bench.check_frames_against_oracle(str(path))
    assert r["tile_size"] == n and r["rows_compared"] == len(rows)
    assert r["variants"]["fp32"]["max_err_over_max_channel"] < 1e-5 and r["variants"]["fp32"]["amplitude_rel_err"] < 1e-6
    assert 0.9e-2 < r["variants"]["broken"]["max_err_over_max_channel"] < 1.1e-2


def test_committed_profile_summaries_match_the_kernels(bench):
    """profiles/traffic.json (PMC-derived HBM bytes per launch) and profiles/kernel_stats.json (rocprofv3 durations)
    must name the kernels bench.py accounts for, and the measured traffic must stay near the algorithmic bytes."""
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert {k + "@2048" for k in bench.KERNEL_BYTES_ACTUAL} <= set(t)
    # the summaries say which kernel sources they were measured with; when those have moved on since, bench.py leaves the figures out
    # of its line (test_stale_profile_summaries_are_not_quoted) -- stale is allowed here, silently stale is not
    assert len(t["_kernel_source_sha16"]) == 16 and all(v.get("kernel_source_sha16") == t["_kernel_source_sha16"] for k, v in t.items() if not k.startswith("_"))
    if t["_kernel_source_sha16"] != bench.kernel_source_sha16():
        import warnings
        warnings.warn("profiles/traffic.json and kernel_stats.json were measured with other kernel sources: regenerate them "
                      "(tools/prof_round.sh, tools/pmc3.sh, tools/profile_summaries.py)")
    n2 = 2048 * 2048
    for k in bench.KERNEL_BYTES_ACTUAL:
        v = t[k + "@2048"]
        own = bench.KERNEL_BYTES_ACTUAL[k] * n2
        assert 0.85 * own <= v["hbm_bytes_per_launch"] <= 1.15 * own, (k, v["hbm_bytes_per_launch"], own)   # no wasted re-reads
        assert os.path.exists(os.path.join(ROOT, v["source"].split(" ")[0]))
    st = json.load(open(os.path.join(ROOT, "profiles", "kernel_stats.json")))
    for k in bench.KERNEL_BYTES_ACTUAL:
        assert st[k + "@2048"]["avg_us"] > 0 and os.path.exists(os.path.join(ROOT, st[k + "@2048"]["source"].split(" ")[0]))


def test_self_launcher_builds_a_torchrun_child_for_n_ranks(bench):
    a = bench.parse(["--gpus", "4", "--steps", "7"])
    cmd = bench.launch_command(a, ["--gpus", "4", "--steps", "7"])
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")


def test_bare_multi_gpu_run_refuses_instead_of_reporting_fewer_gpus():
    """`python bench.py --gpus 2` on a node with fewer GPUs must exit non-zero and print no JSON line
    (round 1 silently ran one rank and printed n_gpus = 1)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("node has >= 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "n_gpus" not in r.stdout
    assert "refusing" in r.stderr


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "n_gpus" not in r.stdout and "refusing" in r.stderr


def test_committed_bench_line_is_reproducible_from_profiles(bench):
    """The roofline fractions of the committed default bench line (profiles/r04n_bench_default.json) can be recomputed from
    the committed rocprofv3 summary (profiles/kernel_stats.json <- r04n_kernel_stats_2048_bench_depth1.csv) and the byte
    accounting of this file: every kernel within 8 % (two processes: the hardware queue a context's stream lands on moves a kernel by up
    to +-1 us, profiles/r03_bimodal_probe.txt section 4a -- 6 % of the 17 us displacement pass), nothing above 1, and the summaries
    regenerate from the CSV."""
    prof = os.path.join(ROOT, "profiles")
    line = [l for l in open(os.path.join(prof, "r04n_bench_default.json")).read().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    st = json.load(open(os.path.join(prof, "kernel_stats.json")))
    r = d["roofline"]
    n2 = 2048 * 2048
    assert d["config"]["tile_size"] == 2048 and d["n_gpus"] == 1 and d["dtype"] == "f32" and d["vs_baseline"] is None
    for k, b in bench.KERNEL_BYTES_ACTUAL.items():
        frac_from_rocprof = b * n2 / (st[k + "@2048"]["avg_us"] * 1e-6) * 1e-9 / bench.HBM_PEAK_GBPS
        assert abs(r["kernels"][k]["frac"] - frac_from_rocprof) <= 0.08 * frac_from_rocprof, (k, r["kernels"][k]["frac"], frac_from_rocprof)
        assert r["kernels"][k]["frac"] < 1.0
    assert r["frac"] == r["kernels"][r["kernel"]]["frac"] and r["frame_frac"] < 1.0 and r["serial_frame_frac"] < 1.0
    assert abs(r["frame_frac"] - bench.FRAME_BYTES_ACTUAL * n2 / (d["ms_per_step"] * 1e-3) * 1e-9 / bench.HBM_PEAK_GBPS) < 1e-9
    # round 4: the timed region is repeated and reported as median with spread; the CPU baseline names the usable cores
    assert d["timing"]["regions"] >= 7 and d["timing"]["ms_per_step_p10"] <= d["ms_per_step"] <= d["timing"]["ms_per_step_p90"]
    assert d["cpu_baseline"]["cores"] <= d["cpu_baseline"]["nproc"] and d["cpu_baseline"]["cores"] == d["cpu_baseline"]["host"]["usable_cpus"]
    assert {"4096x4096_fp32_depth3", "4096x4096_fp16_spectrum_depth3", "4096x4096_fp16_intermediates_depth3"} <= set(d["extra"])
    assert d["extra"]["4096x4096_fp32_depth3"]["error_vs_float64_oracle"] <= 1e-5 < d["extra"]["4096x4096_fp16_spectrum_depth3"]["error_vs_float64_oracle"] <= 1e-3
    # the summary itself regenerates from the committed CSV
    import csv
    import re
    acc = {}
    for row in csv.DictReader(open(os.path.join(prof, "r04n_kernel_stats_2048_bench_depth1.csv"))):
        m = re.search(r"(k_[a-z_0-9]+)<2048", row["Name"])
        name = m.group(1) if m else None
        if name and name.startswith("k_zpass"):          # the z pass's kernel forms (k_zpass, k_zpass_c1) are all the frame's first launch
            name = "k_zpass"
        if name in bench.KERNEL_BYTES_ACTUAL:
            a = acc.setdefault(name, [0, 0.0]); a[0] += int(row["Calls"]); a[1] += float(row["TotalDurationNs"])
    for k, (calls, total) in acc.items():
        assert abs(total / calls * 1e-3 - st[k + "@2048"]["avg_us"]) < 1e-6
