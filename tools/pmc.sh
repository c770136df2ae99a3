#!/bin/bash
# usage: tools/pmc.sh <tag> <N> [frames]  -- separate rocprofv3 --pmc passes (never combined with tracing)
TAG=$1; N=${2:-2048}; FR=${3:-10}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" \
         "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_${TAG}/p$i -- python3 $R/tools/run_frames.py $N $FR > $R/gpurun_out/pmc_${TAG}_p$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${TAG} | tee $R/gpurun_out/pmc_${TAG}_summary.txt
