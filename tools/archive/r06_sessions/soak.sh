# round 6: full GPU suite with the pooled page-locked arrays, then the API soak in its default form (persistent registrations), four seeds
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_pool_pytest.txt 2>&1; echo "pytest exit $?" >> gpurun_out/r06_pool_pytest.txt
tail -3 gpurun_out/r06_pool_pytest.txt
for seed in 101 102 103 104; do
  timeout -k 10 420 python3 tools/soak_api.py 6000 $seed 2>&1 | grep -v amdgpu.ids | tail -2
done > gpurun_out/r06_soak_api.txt 2>&1
cat gpurun_out/r06_soak_api.txt
