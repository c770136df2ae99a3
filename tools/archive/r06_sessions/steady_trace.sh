# round 6 (late): what the steady state of the depth-3 pipeline looks like (kernel trace of bursts of 60 frames; frames 30-47 printed)
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/bt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/bt -- python3 $R/tools/archive/burst_trace.py run 60 3 > $R/gpurun_out/r06_steady_trace.log 2>&1
f=$(find $R/gpurun_out/bt -name "*kernel_trace.csv" | head -1)
python3 $R/tools/archive/burst_trace.py show $f 60 30 18 > $R/gpurun_out/r06_steady_trace.txt 2>&1
rm -rf $R/gpurun_out/bt
cat $R/gpurun_out/r06_steady_trace.txt
