# round 6, GPU session 2: full GPU suite on the refactored library, fault recovery, fused-cmul A/B, the drop-in call, the probe behind the suite
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06_s2_pytest.txt 2>&1; echo "pytest exit $?" >> gpurun_out/r06_s2_pytest.txt
OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_probe.so timeout 300 python tools/slow_window.py 3 2 > gpurun_out/r06_slow_window_after_suite.txt 2>&1
OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_fault.so timeout 300 python tools/fault_probe.py > gpurun_out/r06_s2_fault.txt 2>&1
bash tools/dropin_call.sh 400 > gpurun_out/r06_s2_dropin.txt 2>&1
bash tools/ab_libs.sh "default split" > gpurun_out/r06_s2_cmul_ab.txt 2>&1
tail -5 gpurun_out/r06_s2_pytest.txt; cat gpurun_out/r06_s2_fault.txt gpurun_out/r06_s2_dropin.txt; grep -v amdgpu.ids gpurun_out/r06_s2_cmul_ab.txt | tail -40
