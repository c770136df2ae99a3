for rep in 1 2; do
  for L in dev endwait; do
    export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_$L.so
    for cfg in "2048 8" "2048 1"; do echo -n "[$L pers=0] "; OCEAN_ZPERS=0 python tools/kernel_times.py $cfg 100; done
  done
done
