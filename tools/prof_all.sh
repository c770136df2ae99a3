#!/bin/bash
# usage (on the GPU box, through gpurun): tools/prof_all.sh <tag>
# The whole evidence set of a round's final kernels: tools/prof_round.sh (kernel stats of bench.py at depth 1 / 3, of serial frames at the other sizes,
# the default bench line), the counter passes of serial frames (tools/pmc3.sh) and of the pipelined regime (tools/pmc_depth.sh), the driver's invocation
# of bench.py.  Afterwards, here: tools/prof_all.sh --collect <tag> copies the summaries into profiles/ and rebuilds kernel_stats.json / traffic.json.
if [ "$1" = "--collect" ]; then
  T=$2; cd "$(dirname "$0")/.." || exit 1
  for f in 2048_bench_depth1 2048_bench_depth3 512x1_serial 4096x1_serial 1024x8_serial; do cp gpurun_out/${T}_kernel_stats_$f.csv profiles/; done
  for f in bench_default bench_extra_default bench_steps20_warmup5 bench_extra_steps20_warmup5; do cp gpurun_out/${T}_$f.json profiles/; done
  cp gpurun_out/pmc_${T}_2048_summary.txt profiles/${T}_pmc_2048_single_tile.txt
  cp gpurun_out/pmc_${T}_4096_summary.txt profiles/${T}_pmc_4096_single_tile.txt
  cp gpurun_out/pmc_${T}_1024x8_summary.txt profiles/${T}_pmc_1024_batch8.txt
  cp gpurun_out/pmc_${T}_2048_d3_summary.txt profiles/${T}_pmc_2048_depth3.txt
  cp gpurun_out/pmc_${T}_4096_d3_summary.txt profiles/${T}_pmc_4096_depth3.txt
  python3 tools/profile_summaries.py stats profiles/${T}_kernel_stats_2048_bench_depth1.csv
  python3 tools/profile_summaries.py stats profiles/${T}_kernel_stats_512x1_serial.csv
  python3 tools/profile_summaries.py stats profiles/${T}_kernel_stats_4096x1_serial.csv
  python3 tools/profile_summaries.py stats profiles/${T}_kernel_stats_1024x8_serial.csv 8
  python3 tools/profile_summaries.py traffic profiles/${T}_pmc_2048_single_tile.txt 2048
  python3 tools/profile_summaries.py traffic profiles/${T}_pmc_4096_single_tile.txt 4096
  python3 tools/profile_summaries.py traffic profiles/${T}_pmc_1024_batch8.txt 1024 8
  python3 tools/profile_summaries.py traffic profiles/${T}_pmc_2048_depth3.txt 2048 1 2.0 3
  python3 tools/profile_summaries.py traffic profiles/${T}_pmc_4096_depth3.txt 4096 1 2.0 3
  exit 0
fi
T=$1
bash tools/prof_round.sh $T
bash tools/pmc3.sh ${T}_2048 2048 40 1 > /dev/null
bash tools/pmc3.sh ${T}_4096 4096 20 1 > /dev/null
bash tools/pmc3.sh ${T}_1024x8 1024 30 8 > /dev/null
bash tools/pmc_depth.sh ${T}_2048_d3 2048 60 1 3 > /dev/null
bash tools/pmc_depth.sh ${T}_4096_d3 4096 30 1 3 > /dev/null
cd $GRAFT_REPO_ROOT && python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_steps20_warmup5.json 2> gpurun_out/${T}_bench_steps20_warmup5.err
echo steps20 rc=$?
cp bench_extra.json gpurun_out/${T}_bench_extra_steps20_warmup5.json
ls gpurun_out | grep $T | head -40
