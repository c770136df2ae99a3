# round 6, GPU session 3: D2H paths, fault recovery (all cases), sincos / butterfly ablations of the z pass, the tuning rules' A/B
mkdir -p gpurun_out
./tools/ubench/d2h > gpurun_out/r06_s3_d2h.txt 2>&1
OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_fault.so timeout 300 python tools/fault_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_s3_fault.txt
for rep in 1 2 3; do for L in default nosincos nofft; do
  if [ "$L" = "default" ]; then unset OCEAN_HIP_LIB; else export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_$L.so; fi
  echo "[$L] $(python tools/kernel_times.py 2048 1 1000 2>&1 | grep -v amdgpu.ids)"
done; done > gpurun_out/r06_s3_ablations.txt 2>&1
unset OCEAN_HIP_LIB
timeout 1200 python tools/ab_tuning.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_s3_tuning.txt
cat gpurun_out/r06_s3_d2h.txt gpurun_out/r06_s3_fault.txt gpurun_out/r06_s3_ablations.txt gpurun_out/r06_s3_tuning.txt
