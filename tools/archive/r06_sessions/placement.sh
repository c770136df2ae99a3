# round 6: ocean_prepare's placement search -- its GPU test, the full suite's duration with it, what it buys (tools/placement_probe.py)
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_api_state_gpu.py -m gpu -x -q -k "placement or compute_waves_read" > gpurun_out/r06_place_test.txt 2>&1; echo "test exit $?" >> gpurun_out/r06_place_test.txt
timeout -k 5 200 python tools/placement_probe.py 2048 14 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_place_probe.txt
timeout -k 5 200 python tools/placement_probe.py 4096 6 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_place_probe.txt
timeout -k 10 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r06_place_pytest.txt 2>&1; echo "pytest exit $?" >> gpurun_out/r06_place_pytest.txt
tail -4 gpurun_out/r06_place_test.txt; cat gpurun_out/r06_place_probe.txt; grep -n "passed\|failed" gpurun_out/r06_place_pytest.txt | tail -2
