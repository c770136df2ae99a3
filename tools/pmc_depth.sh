#!/bin/bash
# usage: tools/pmc_depth.sh <tag> <N> <frames> <tiles> <depth>
# HBM-side counters of the PIPELINED regime (bench.py's headline: depth 3, maps stored non-temporally, three chains rotating through the
# memory-side cache): FETCH_SIZE and WRITE_SIZE in passes of their own (MI355X_MICROARCH.md: they do not fit one pass), the L2 hit / miss and
# fabric request counters beside them.  Counter passes only -- never combined with a trace.
TAG=$1; N=$2; FR=$3; TL=$4; DP=$5
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
         "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" \
         "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_${TAG}/p$i -- python3 $R/tools/run_frames.py $N $FR $TL $DP > $R/gpurun_out/pmc_${TAG}_p$i.log 2>&1 || tail -3 $R/gpurun_out/pmc_${TAG}_p$i.log
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${TAG} | tee $R/gpurun_out/pmc_${TAG}_summary.txt
