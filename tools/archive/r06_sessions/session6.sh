# round 6, GPU session 6: which of the two z-pass changes costs what (cycles, per-XCC busy time), does the slow XCD follow the column group, full suite
mkdir -p gpurun_out
for rep in 1 2; do for L in probe probe_pt probe_tt probe_r05z; do
  echo "##### $L"; OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_$L.so timeout 300 python tools/slow_window.py 0 2>&1 | grep -E "^==|duration us|kilocycles|busy time"
done; done > gpurun_out/r06_s6_cycles.txt 2>&1
echo "##### rotation sweep (probe_r05z)" >> gpurun_out/r06_s6_cycles.txt
OCEAN_XCD_ROT_SWEEP=1 OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_probe_r05z.so timeout 300 python tools/slow_window.py 0 2>&1 | grep -E "^==|duration us|kilocycles|busy time|per-XCC median" >> gpurun_out/r06_s6_cycles.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06_s6_pytest.txt 2>&1; echo "pytest exit $?" >> gpurun_out/r06_s6_pytest.txt
tail -4 gpurun_out/r06_s6_pytest.txt; cat gpurun_out/r06_s6_cycles.txt
