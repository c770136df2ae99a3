# round 6, GPU session 8 (strict timeouts): full suite on the final kernels; the drop-in call with direct host stores; the ramp rule against
# rounds 4-5's constants (developer build, env overrides); where the workgroups run and whether the slow XCD follows the column group
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_s8_pytest.txt 2>&1; echo "pytest exit $?" >> gpurun_out/r06_s8_pytest.txt
timeout -k 5 200 bash tools/dropin_call.sh 400 > gpurun_out/r06_s8_dropin.txt 2>&1
export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_dev.so
for rep in 1 2 3 4; do
  echo "[rule]   $(timeout -k 5 90 python tools/kernel_times.py 2048 1 1000 2>&1 | grep -v amdgpu.ids)"
  echo "[r05]    $(OCEAN_RAMP_Z=500 OCEAN_RAMP_B=450 OCEAN_RAMP_D=450 timeout -k 5 90 python tools/kernel_times.py 2048 1 1000 2>&1 | grep -v amdgpu.ids)"
  echo "[rule]   $(OCEAN_FRAMES=3000 timeout -k 5 90 python tools/depth_batch.py 2048 1 3 2>&1 | grep -v amdgpu.ids)"
  echo "[r05]    $(OCEAN_FRAMES=3000 OCEAN_RAMP_Z=500 OCEAN_RAMP_B=900 OCEAN_RAMP_D=900 timeout -k 5 90 python tools/depth_batch.py 2048 1 3 2>&1 | grep -v amdgpu.ids)"
done > gpurun_out/r06_s8_ramp.txt 2>&1
export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_probe.so
OCEAN_XCD_ROT_SWEEP=1 timeout -k 5 150 python tools/slow_window.py 0 2>&1 | grep -E "^==|duration us|kilocycles|busy time|per-XCC median|workgroups per" > gpurun_out/r06_s8_placement.txt
unset OCEAN_HIP_LIB
tail -4 gpurun_out/r06_s8_pytest.txt; cat gpurun_out/r06_s8_dropin.txt gpurun_out/r06_s8_ramp.txt gpurun_out/r06_s8_placement.txt
