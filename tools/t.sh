R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for N in 512 1024; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s$N -- python3 $R/tools/run_frames.py $N 200 1 > /dev/null 2>&1
find $R/gpurun_out/prof_s$N -name "*kernel_stats.csv" -exec head -4 {} \;
done
