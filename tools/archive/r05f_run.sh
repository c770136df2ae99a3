mkdir -p gpurun_out/r05f
setlib() { if [ "$1" = "default" ]; then unset OCEAN_HIP_LIB; else export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_$1.so; fi; }
for rep in 1 2 3; do for L in default ntall nont; do setlib $L; echo "[$L] $(python tools/kernel_times.py 4096 1 60)"; echo "[$L] $(python tools/depth_batch.py 4096 1 2)"; echo "[$L] $(python tools/depth_batch.py 4096 1 3)"; done; done 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee gpurun_out/r05f/ab_nt.log
unset OCEAN_HIP_LIB
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r05f/gputests.log 2>&1; echo tests rc=$?; tail -3 gpurun_out/r05f/gputests.log
