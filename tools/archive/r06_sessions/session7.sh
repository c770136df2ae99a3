# round 6, GPU session 7 (strict timeouts): z pass with phase + twiddle tables (tables built from registers behind the loads) against the round-5 kernel; full suite
mkdir -p gpurun_out
for rep in 1 2 3 4; do for L in default r05z; do
  if [ "$L" = "default" ]; then unset OCEAN_HIP_LIB; else export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_$L.so; fi
  for cfg in "2048 1 1000" "4096 1 100" "1024 8 300"; do echo "[$L] $(timeout -k 5 90 python tools/kernel_times.py $cfg 2>&1 | grep -v amdgpu.ids)"; done
done; done > gpurun_out/r06_s7_times.txt 2>&1
unset OCEAN_HIP_LIB
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_s7_pytest.txt 2>&1; echo "pytest exit $?" >> gpurun_out/r06_s7_pytest.txt
timeout -k 5 200 bash tools/dropin_call.sh 400 > gpurun_out/r06_s7_dropin.txt 2>&1
tail -4 gpurun_out/r06_s7_pytest.txt; cat gpurun_out/r06_s7_times.txt gpurun_out/r06_s7_dropin.txt
