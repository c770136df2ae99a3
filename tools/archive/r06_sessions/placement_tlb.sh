# round 6: translation-cache counters of the z pass by candidate placement (tools/placement_tlb.py)
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "utcl|tlb|translat|UTCL2|TA_BUSY" | cut -c1-200 | sort -u | head -60 > $R/gpurun_out/r06_tlb_counters.txt
i=0
for C in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/tlb/p$i -- python3 $R/tools/placement_tlb.py > $R/gpurun_out/r06_tlb_p$i.log 2>&1
  f=$(find $R/gpurun_out/tlb/p$i -name "*counter_collection.csv" | head -1)
  { echo "## pass $i: $C"; grep -E "placement search|report" $R/gpurun_out/r06_tlb_p$i.log; python3 $R/tools/placement_tlb.py --summarise $f; } >> $R/gpurun_out/r06_tlb_summary.txt 2>&1
done
cd $R; rm -rf gpurun_out/tlb
cat gpurun_out/r06_tlb_counters.txt | head -30; cat gpurun_out/r06_tlb_summary.txt
