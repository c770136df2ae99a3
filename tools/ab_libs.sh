# usage: ab_libs.sh "<libs>" : hashes + serial kernel times + pipelined frames for variant libraries (default = the shipped one), interleaved
LIBS="$1"
setlib() { if [ "$1" = "default" ]; then unset OCEAN_HIP_LIB; else export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_$1.so; fi; }
for L in $LIBS; do setlib $L; for cfg in "512 1" "1024 8" "2048 1" "2048 1 3" "4096 1" "2048 1 1 3" "2048 1 1 0 16" "256 4"; do echo "[$L] $(python tools/frame_hash.py $cfg)"; done; done
for rep in 1 2 3; do
  for L in $LIBS; do setlib $L
    for cfg in "2048 1" "1024 8" "512 1"; do echo "[$L] $(python tools/kernel_times.py $cfg)"; done
    echo "[$L] $(python tools/kernel_times.py 4096 1 60)"
    echo "[$L] $(python tools/depth_batch.py 2048 1 3)"
  done
done
