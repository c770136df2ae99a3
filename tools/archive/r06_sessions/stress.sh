# round 6, final library: the older stress / soak tools once more (regressions of the round's host-side changes would show here)
mkdir -p gpurun_out
{
echo "== stress_modes"; timeout -k 10 400 python3 tools/stress_modes.py 2>&1 | grep -v amdgpu.ids | tail -3
echo "== stress_tiles"; timeout -k 10 400 python3 tools/stress_tiles.py 2>&1 | grep -v amdgpu.ids | tail -3
echo "== soak_merged 90 s"; timeout -k 10 300 python3 tools/soak_merged.py 90 7 2>&1 | grep -v amdgpu.ids | tail -3
echo "== soak"; timeout -k 10 400 python3 tools/soak.py 2>&1 | grep -v amdgpu.ids | tail -4
} > gpurun_out/r06_stress.txt 2>&1
cat gpurun_out/r06_stress.txt
