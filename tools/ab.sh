# usage: ab.sh "<libs>" N tiles depths   (developer A/B across variant libraries, 3 interleaved repeats)
for rep in 1 2 3; do
  for L in $1; do
    if [ "$L" = "default" ]; then unset OCEAN_HIP_LIB; else export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_$L.so; fi
    echo -n "[$L] "; python tools/depth_batch.py $2 $3 $4
  done
done
