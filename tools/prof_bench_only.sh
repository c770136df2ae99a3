R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for D in 1 3; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02v_d$D -- python3 $R/bench.py --depth $D --no-cpu-baseline --no-extra --no-gather > $R/gpurun_out/prof_r02v_d$D.json 2> $R/gpurun_out/prof_r02v_d$D.err
find $R/gpurun_out/prof_r02v_d$D -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r02v_kernel_stats_2048_bench_depth$D.csv \;
done
cd $R && python3 bench.py > gpurun_out/r02v_bench_default.json 2> gpurun_out/r02v_bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02v_bench_steps20.json 2>/dev/null
