#!/bin/bash
# usage: tools/pmc2.sh <tag> <N> <frames> "<group1>" "<group2>" ...   (each group = one rocprofv3 --pmc pass)
TAG=$1; N=$2; FR=$3; shift 3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_${TAG}/p$i -- python3 $R/tools/run_frames.py $N $FR > $R/gpurun_out/pmc_${TAG}_p$i.log 2>&1 || tail -5 $R/gpurun_out/pmc_${TAG}_p$i.log
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${TAG} | tee $R/gpurun_out/pmc_${TAG}_summary.txt
