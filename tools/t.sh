bash tools/ab.sh "default z6 xb4 z6xb4" 2048 1 1,2,3
for L in default z6 xb4; do
  if [ "$L" = "default" ]; then unset OCEAN_HIP_LIB; else export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_$L.so; fi
  echo -n "[$L] "; python tools/quick_bench.py 2048 | grep -o "kernels.*"
done
