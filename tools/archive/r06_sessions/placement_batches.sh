# round 6: does the allocation matter for batches of small tiles too (the search's size rule)?
mkdir -p gpurun_out
{
echo "##### 512^2 x 16"; timeout -k 10 300 python3 tools/placement_probe.py 512 12 16 6 2>&1 | grep -v amdgpu.ids
echo "##### 1024^2 x 8"; timeout -k 10 300 python3 tools/placement_probe.py 1024 12 8 6 2>&1 | grep -v amdgpu.ids
echo "##### 256^2 x 64"; timeout -k 10 300 python3 tools/placement_probe.py 256 12 64 6 2>&1 | grep -v amdgpu.ids
echo "##### 1024^2 x 1"; timeout -k 10 300 python3 tools/placement_probe.py 1024 12 1 6 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r06_place_batches.txt 2>&1
grep -E "#####|^search" gpurun_out/r06_place_batches.txt
