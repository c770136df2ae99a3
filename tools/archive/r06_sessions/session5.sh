# round 6, GPU session 5: where do the workgroups run (placement per launch), cycles per launch old / new kernel, then the full GPU suite
mkdir -p gpurun_out
for rep in 1 2 3; do for L in probe probe_r05z; do
  echo "##### $L"; OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_$L.so timeout 300 python tools/slow_window.py 0 2>&1 | grep -E "^==|duration us|clock GHz|kilocycles|workgroups per|busy time"
done; done > gpurun_out/r06_s5_cycles.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06_s5_pytest.txt 2>&1; echo "pytest exit $?" >> gpurun_out/r06_s5_pytest.txt
for rep in 1 2 3; do for L in default r05z; do
  if [ "$L" = "default" ]; then unset OCEAN_HIP_LIB; else export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_$L.so; fi
  for cfg in "2048 1 1000" "4096 1 100"; do echo "[$L] $(python tools/kernel_times.py $cfg 2>&1 | grep -v amdgpu.ids)"; done
done; done > gpurun_out/r06_s5_times.txt 2>&1
tail -4 gpurun_out/r06_s5_pytest.txt; cat gpurun_out/r06_s5_cycles.txt gpurun_out/r06_s5_times.txt
