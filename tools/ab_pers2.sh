export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_dev.so
for rep in 1 2; do
  for cfg in "2048 8" "2048 3" "1024 32"; do
    for P in 0 1; do echo -n "[pers=$P] "; OCEAN_ZPERS=$P python tools/kernel_times.py $cfg 100; done
  done
done
