export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_dev.so
for cfg in "4096 1" "4096 1 1 3" "4096 1 1 0 16" "4096 1 2" "4096 1 1 1" "4096 1 1 2"; do
  for P in 0 1; do OCEAN_ZC1=$P python tools/frame_hash.py $cfg; done
done
for rep in 1 2; do
  for P in 0 1; do echo -n "[c1=$P] "; OCEAN_ZC1=$P python tools/kernel_times.py 4096 1 60; done
done
for rep in 1 2; do
  for P in 0 1; do echo -n "[c1=$P] "; OCEAN_FRAMES=300 OCEAN_WARMUP=150 OCEAN_ZC1=$P python tools/depth_batch.py 4096 1 3,2 2>&1 | tail -1; done
done
