# round 6 (late): the z pass's staggered start re-swept on the final kernel (developer build: OCEAN_RAMP_Z / _B / _D in 10 ns ticks; the rule gives 474 / 1210+ / ...)
mkdir -p gpurun_out
export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_dev.so
{
for rep in 1 2 3; do
  for z in rule 0 250 350 474 600 750 950; do
    if [ $z = rule ]; then unset OCEAN_RAMP_Z; else export OCEAN_RAMP_Z=$z; fi
    echo -n "ramp_z=$z  "; timeout -k 5 100 python3 tools/kernel_times.py 2048 1 1000 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
unset OCEAN_RAMP_Z
for rep in 1 2; do
  for b in rule 0 400 800 1200 1600; do
    if [ $b = rule ]; then unset OCEAN_RAMP_B; else export OCEAN_RAMP_B=$b; fi
    echo -n "ramp_b=$b  "; timeout -k 5 100 python3 tools/kernel_times.py 2048 1 1000 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
unset OCEAN_RAMP_B
for rep in 1 2; do
  for d in rule 0 400 800 1200 1600; do
    if [ $d = rule ]; then unset OCEAN_RAMP_D; else export OCEAN_RAMP_D=$d; fi
    echo -n "ramp_d=$d  "; timeout -k 5 100 python3 tools/kernel_times.py 2048 1 1000 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
} > gpurun_out/r06_ramp_sweep.txt 2>&1
cat gpurun_out/r06_ramp_sweep.txt
