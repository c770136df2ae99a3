mkdir -p gpurun_out/r05e
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_variants_gpu.py -m gpu -x -q -k "2048 or 4096 or every_selectable or merged" > gpurun_out/r05e/tests.log 2>&1; echo tests rc=$?; tail -3 gpurun_out/r05e/tests.log
setlib() { if [ "$1" = "default" ]; then unset OCEAN_HIP_LIB; else export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_$1.so; fi; }
for rep in 1 2 3; do for L in default full; do setlib $L; echo "[$L] $(python tools/kernel_times.py 2048 1)"; echo "[$L] $(python tools/kernel_times.py 4096 1 60)"; echo "[$L] $(python tools/depth_batch.py 2048 1 3)"; echo "[$L] $(python tools/depth_batch.py 4096 1 3)"; done; done 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee gpurun_out/r05e/ab_half.log
unset OCEAN_HIP_LIB
for cfg in "128 1 1" "512 1 4" "512 1 2" "256 1 4"; do python tools/ab_xmerge.py $cfg 2000 3; done 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee gpurun_out/r05e/ab_xmerge2.log
g++ -O2 -std=c++17 tools/ubench/sync_tail.cpp -Iinclude -Lwatersurfacerendering_amd -locean_hip -Wl,-rpath,$PWD/watersurfacerendering_amd -o /tmp/sync_tail && for cfg in "2048 10000 -1" "2048 10000 3" "512 10000 -1" "512 10000 3" "2048 10000 -1"; do /tmp/sync_tail $cfg; done 2>&1 | tee gpurun_out/r05e/sync_tail.log
python tools/sync_cost.py 512,2048 2>&1 | grep "N=" | tee -a gpurun_out/r05e/sync_tail.log
