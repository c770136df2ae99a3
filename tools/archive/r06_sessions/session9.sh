# round 6, GPU session 9 (strict timeouts): wave-local z-pass transforms (WaveFFT) -- parity, A/B against the engine plan, full suite
mkdir -p gpurun_out
{ timeout -k 5 120 python tools/parity_one.py 2048; timeout -k 5 200 python tools/parity_one.py 4096; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_s9_parity.txt
for rep in 1 2 3 4; do for L in default nowf; do
  if [ "$L" = "default" ]; then unset OCEAN_HIP_LIB; else export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_$L.so; fi
  for cfg in "2048 1 1000" "4096 1 100"; do echo "[$L] $(timeout -k 5 90 python tools/kernel_times.py $cfg 2>&1 | grep -v amdgpu.ids)"; done
  echo "[$L] $(OCEAN_FRAMES=2000 timeout -k 5 90 python tools/depth_batch.py 2048 1 3 2>&1 | grep -v amdgpu.ids)"
done; done > gpurun_out/r06_s9_times.txt 2>&1
unset OCEAN_HIP_LIB
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_s9_pytest.txt 2>&1; echo "pytest exit $?" >> gpurun_out/r06_s9_pytest.txt
cat gpurun_out/r06_s9_parity.txt gpurun_out/r06_s9_times.txt; tail -4 gpurun_out/r06_s9_pytest.txt
