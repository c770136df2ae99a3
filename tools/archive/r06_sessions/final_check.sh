# round 6, final check of the committed state: full GPU suite, the fault-injection probe, bench.py as the driver runs it
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_final_pytest.txt 2>&1; echo "pytest exit $?" >> gpurun_out/r06_final_pytest.txt
OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_fault.so timeout -k 5 200 python tools/fault_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_final_fault.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"SMOKE_OK\")" > gpurun_out/r06_final_smoke.txt 2>&1; tail -1 gpurun_out/r06_final_smoke.txt
timeout -k 10 900 python bench.py > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err; echo "bench rc $?"
cp bench_extra.json gpurun_out/r06_final_bench_extra.json
tail -3 gpurun_out/r06_final_pytest.txt; cat gpurun_out/r06_final_fault.txt; cat gpurun_out/r06_final_bench.json
