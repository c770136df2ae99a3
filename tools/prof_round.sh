#!/bin/bash
# usage: tools/prof_round.sh <tag>   -- rocprofv3 kernel stats of bench.py (serial + default depth) and the plain bench JSON
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for D in 1 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_d$D -- python3 $R/bench.py --depth $D --no-cpu-baseline --no-extra > $R/gpurun_out/prof_${TAG}_d$D.json 2> $R/gpurun_out/prof_${TAG}_d$D.err
  find $R/gpurun_out/prof_${TAG}_d$D -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_kernel_stats_2048_bench_depth$D.csv \;
done
cd $R && python3 bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
