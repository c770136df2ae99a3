#!/bin/bash
# usage: tools/prof_round.sh <tag>
# rocprofv3 evidence of one round (run on the GPU box through gpurun; outputs under gpurun_out/, copy what is judged into profiles/):
#   kernel stats of bench.py itself (serial frames = the regime of its roofline object, and the default depth),
#   kernel stats of serial frames at 512^2, 4096^2 and 8 x 1024^2, and the plain default bench JSON.
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for D in 1 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_d$D -- python3 $R/bench.py --depth $D --steps 600 --warmup 300 --no-cpu-baseline --no-extra --no-gather > $R/gpurun_out/prof_${TAG}_d$D.json 2> $R/gpurun_out/prof_${TAG}_d$D.err
  find $R/gpurun_out/prof_${TAG}_d$D -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_kernel_stats_2048_bench_depth$D.csv \;
done
for CFG in "512 3000 1" "4096 400 1" "1024 600 8"; do
  set -- $CFG
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_n$1x$3 -- python3 $R/tools/run_frames.py $1 $2 $3 > $R/gpurun_out/prof_${TAG}_n$1x$3.log 2>&1
  find $R/gpurun_out/prof_${TAG}_n$1x$3 -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_kernel_stats_$1x$3_serial.csv \;
done
rocprofv3 -L 2>/dev/null | grep -i -B1 -A4 "RDREQ\|WRREQ" > $R/gpurun_out/${TAG}_counters_rdreq.txt
cd $R && python3 bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
echo bench rc=$?
cp bench_extra.json gpurun_out/${TAG}_bench_extra_default.json      # the sidecar of that run (everything beside the compact stdout line)
