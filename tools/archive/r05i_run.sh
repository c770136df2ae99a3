mkdir -p gpurun_out/r05i
setlib() { if [ "$1" = "default" ]; then unset OCEAN_HIP_LIB; else export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_$1.so; fi; }
for rep in 1 2 3 4 5 6 7 8; do for L in default nowt; do setlib $L; echo "[$L] $(python tools/depth_batch.py 2048 1 3)"; done; done 2>&1 | grep "d3=" | tee gpurun_out/r05i/d3.log
for rep in 1 2 3; do for L in default nowt; do setlib $L; echo "[$L] $(python tools/depth_batch.py 2048 1 2)"; echo "[$L] $(python tools/depth_batch.py 1024 8 2)"; echo "[$L] $(python tools/depth_batch.py 512 16 2)"; done; done 2>&1 | grep "us" | tee gpurun_out/r05i/other.log
