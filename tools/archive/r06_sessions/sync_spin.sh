# round 6: ocean_synchronize polled-then-blocking (OceanTuning::sync_spin_us) against the plain blocking drain, bursts as bench.py --steps 20 times them
mkdir -p gpurun_out
{
for rep in 1 2; do
  echo "##### spin (shipped)"; timeout -k 10 200 python3 tools/burst_probe.py 3 2>&1 | grep -v amdgpu.ids
  echo "##### nospin"; OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_nospin.so timeout -k 10 200 python3 tools/burst_probe.py 3 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r06_sync_spin.txt 2>&1
cat gpurun_out/r06_sync_spin.txt
