export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_dev.so
for cfg in "2048 1" "2048 1 3" "1024 1" "1024 8" "1024 8 2" "2048 1 1 3" "2048 1 1 0 16"; do
  for P in 0 1; do OCEAN_ZC1=$P python tools/frame_hash.py $cfg; done
done
for rep in 1 2; do
  for cfg in "2048 1" "1024 8" "1024 1"; do
    for P in 0 1; do echo -n "[c1=$P] "; OCEAN_ZC1=$P python tools/kernel_times.py $cfg; done
  done
done
for rep in 1 2 3; do
  for P in 0 1; do echo -n "[c1=$P] "; OCEAN_ZC1=$P python tools/depth_batch.py 2048 1 3,1 2>&1 | tail -1; done
done
for rep in 1 2; do
  for P in 0 1; do echo -n "[c1=$P] "; OCEAN_FRAMES=400 OCEAN_WARMUP=200 OCEAN_ZC1=$P python tools/depth_batch.py 1024 8 2,1 2>&1 | tail -1; done
done
