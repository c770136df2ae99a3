# ZW=2 with kz in registers (dev) against the library built before it (default)
for L in default dev; do
  if [ "$L" = "default" ]; then unset OCEAN_HIP_LIB; else export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_$L.so; fi
  python tools/frame_hash.py 1024 8 2; python tools/frame_hash.py 2048 1 3 3; python tools/frame_hash.py 4096 1 2; python tools/frame_hash.py 2048 1 4 0 16
done
for rep in 1 2 3; do
  for L in default dev; do
    if [ "$L" = "default" ]; then unset OCEAN_HIP_LIB; else export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_$L.so; fi
    echo -n "[$L] "; OCEAN_FRAMES=400 OCEAN_WARMUP=200 python tools/depth_batch.py 1024 8 2 | tail -1
    echo -n "[$L jac] "; OCEAN_MODE=3 python tools/depth_batch.py 2048 1 3 | tail -1
    echo -n "[$L] "; python tools/depth_batch.py 2048 1 4 | tail -1
    echo -n "[$L] "; OCEAN_FRAMES=300 OCEAN_WARMUP=150 python tools/depth_batch.py 4096 1 3 | tail -1
  done
done
