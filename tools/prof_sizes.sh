#!/bin/bash
# usage: tools/prof_sizes.sh <tag>   -- rocprofv3 kernel stats of bench.py at the other BASELINE sizes (512^2, 4096^2)
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for N in 512 4096; do
  if [ $N = 4096 ]; then EXTRA="--steps 400 --warmup 200"; else EXTRA=""; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_n$N -- python3 $R/bench.py --size $N $EXTRA --no-cpu-baseline --no-extra > $R/gpurun_out/prof_${TAG}_n$N.json 2> $R/gpurun_out/prof_${TAG}_n$N.err
  find $R/gpurun_out/prof_${TAG}_n$N -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_kernel_stats_${N}_bench_depth3.csv \;
done
