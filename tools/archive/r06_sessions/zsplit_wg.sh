# round 6 (late): small serial frames -- a column's four transforms shared by two workgroups of ONE launch (developer build, OCEAN_ZSPLIT_WG=1)
mkdir -p gpurun_out
export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_zs.so
{
for n in 128 512; do OCEAN_ZSPLIT_WG=1 timeout -k 5 100 python3 tools/parity_one.py $n 2>&1 | grep -v amdgpu.ids | head -3; done
for rep in 1 2 3; do
  for n in 64 128 256 512; do
    for sw in 0 1; do
      echo -n "split=$sw  "; OCEAN_ZSPLIT_WG=$sw timeout -k 5 100 python3 tools/kernel_times.py $n 1 2000 2>&1 | grep -v amdgpu.ids | tail -1
    done
  done
done
} > gpurun_out/r06_zsplit_wg.txt 2>&1
cat gpurun_out/r06_zsplit_wg.txt
