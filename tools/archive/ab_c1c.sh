export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_dev.so
for rep in 1 2 3 4 5; do
  for P in 0 1; do echo "[c1=$P] $(OCEAN_FRAMES=600 OCEAN_WARMUP=300 OCEAN_ZC1=$P python tools/depth_batch.py 4096 1 3,2 | tail -1)"; done
done
for rep in 1 2 3; do
  for P in 0 1; do echo "[c1=$P z16] $(OCEAN_Z16=1 OCEAN_FRAMES=600 OCEAN_WARMUP=300 OCEAN_ZC1=$P python tools/depth_batch.py 4096 1 3 | tail -1)"; done
done
for T in 2 3 4; do for P in 0 1; do echo "[c1=$P] $(OCEAN_ZC1=$P python tools/kernel_times.py 1024 $T)"; done; done
for P in 0 1; do echo "[c1=$P] $(OCEAN_ZC1=$P python tools/kernel_times.py 2048 4 100)"; done
