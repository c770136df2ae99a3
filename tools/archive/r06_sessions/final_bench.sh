# the committed bench lines of round 6: run AFTER profiles/kernel_stats.json / traffic.json were regenerated for the final kernels, so that the lines quote them
mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/r06n_bench_default.json 2> gpurun_out/r06n_bench_default.err; echo "bench rc $?"
cp bench_extra.json gpurun_out/r06n_bench_extra_default.json
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06n_bench_steps20_warmup5.json 2> gpurun_out/r06n_bench_steps20_warmup5.err; echo "steps20 rc $?"
cp bench_extra.json gpurun_out/r06n_bench_extra_steps20_warmup5.json
python - <<'PY'
import json
for f in ("r06n_bench_default", "r06n_bench_steps20_warmup5"):
    d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][0]); r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["kernel"], r["launch_us"], r["frac"], r["rocprof_launch_us"], r["profiles_current"], {k: round(v["launch_us"], 2) for k, v in r["kernels"].items()})
PY
