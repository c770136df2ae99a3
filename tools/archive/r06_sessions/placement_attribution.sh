cd /root/repo
{
timeout -k 5 300 python3 tools/placement_attribution.py 2048 1 12 3
timeout -k 5 200 python3 tools/placement_attribution.py 1024 8 8 2
timeout -k 5 200 python3 tools/placement_attribution.py 4096 1 6 2
} > gpurun_out/r06_place_attr.txt 2>&1
timeout -k 5 400 python3 -m pytest tests/test_api_state_gpu.py -x -q -m gpu > gpurun_out/r06_place_attr_pytest.txt 2>&1
tail -3 gpurun_out/r06_place_attr_pytest.txt
