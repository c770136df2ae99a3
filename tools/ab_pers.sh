export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_dev.so
for cfg in "1024 1" "1024 8" "2048 1" "2048 1 1 3" "2048 1 1 0 16" "1024 8 2"; do
  for P in 0 1; do OCEAN_ZPERS=$P python tools/frame_hash.py $cfg; done
done
for rep in 1 2; do
  for cfg in "1024 8" "2048 1" "1024 1"; do
    for P in 0 1; do echo -n "[pers=$P] "; OCEAN_ZPERS=$P python tools/kernel_times.py $cfg; done
  done
done
for G in 512 640 768 1025; do echo -n "[pers grid $G] "; OCEAN_ZPERS=1 OCEAN_ZPERS_GRID=$G python tools/kernel_times.py 2048 1; done
