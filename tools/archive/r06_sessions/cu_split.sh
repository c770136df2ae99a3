# round 6 (late): pipelined z passes on a stream of their own, confined to k CUs of every XCD (developer build, OCEAN_CU_SPLIT=k; 0 = the stream
# structure without masks; OCEAN_CU_SPLIT_XALL=1: only the z stream is confined).  us per frame at 2048^2, depths 2 / 3 / 4.
mkdir -p gpurun_out
export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_cus.so OCEAN_FRAMES=3000 OCEAN_WARMUP=1500
{
echo "== correctness under the split (k = 12): last frame of a long pipelined run against a fresh context"
OCEAN_CU_SPLIT=12 timeout -k 10 200 python3 tools/soak.py 2>&1 | grep -v amdgpu.ids | head -1
for rep in 1 2; do
  unset OCEAN_CU_SPLIT OCEAN_CU_SPLIT_XALL
  echo -n "shipped structure     "; timeout -k 5 120 python3 tools/depth_batch.py 2048 1 2,3,4 2>&1 | grep -v amdgpu.ids | tail -1
  for k in 0 8 12 16 20 24; do
    export OCEAN_CU_SPLIT=$k; unset OCEAN_CU_SPLIT_XALL
    echo -n "z on $k CUs/XCD, x on rest  "; timeout -k 5 120 python3 tools/depth_batch.py 2048 1 2,3,4 2>&1 | grep -v amdgpu.ids | tail -1
  done
  for k in 12 16 20 24; do
    export OCEAN_CU_SPLIT=$k OCEAN_CU_SPLIT_XALL=1
    echo -n "z on $k CUs/XCD, x anywhere "; timeout -k 5 120 python3 tools/depth_batch.py 2048 1 2,3,4 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
} > gpurun_out/r06_cu_split.txt 2>&1
cat gpurun_out/r06_cu_split.txt
