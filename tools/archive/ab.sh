# usage: ab.sh "<libs>" N tiles depths   (developer A/B across variant libraries, 3 interleaved repeats)
# Steady state needs a few hundred frames: defaults are 500 warm-up + 1000 timed frames (OCEAN_WARMUP / OCEAN_FRAMES).
for rep in 1 2 3; do
  for L in $1; do
    if [ "$L" = "default" ]; then unset OCEAN_HIP_LIB; else export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_$L.so; fi
    echo -n "[$L] "; python tools/depth_batch.py $2 $3 $4
  done
done
