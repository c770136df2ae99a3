# round 6, GPU session 1: the slow window's telemetry + the radix-16 single-transform z pass (A/B)
mkdir -p gpurun_out
export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_probe.so
timeout 900 python tools/slow_window.py 120 4 > gpurun_out/r06_slow_window_raw.txt 2>&1
unset OCEAN_HIP_LIB
{
  OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_r16.so python tools/parity_one.py 2048
  python tools/parity_one.py 2048
  bash tools/ab_kernels.sh "default r16 default r16" 2048
  for L in default r16; do
    if [ "$L" = "default" ]; then unset OCEAN_HIP_LIB; else export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_$L.so; fi
    echo "[$L] $(python tools/depth_batch.py 2048 1 1,3)"
  done
} > gpurun_out/r06_r16_ab.txt 2>&1
tail -5 gpurun_out/r06_slow_window_raw.txt; cat gpurun_out/r06_r16_ab.txt
