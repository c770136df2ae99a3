#!/bin/bash
# The reference's call shape through the C++ adaptor (include/WSTessendorf.hpp): blocking ComputeWaves = synthesis + both maps in host memory,
# us per call and achieved PCIe GB/s at 512^2 / 1024^2 / 2048^2 (tests/cpp/adaptor_demo.cpp prints them on stderr).  usage: tools/dropin_call.sh [frames]
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
g++ -std=c++17 -O2 -I include tests/cpp/adaptor_demo.cpp -o gpurun_out/adaptor_demo -L watersurfacerendering_amd -locean_hip \
    -Wl,-rpath,"$PWD/watersurfacerendering_amd" -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib || exit 1
for rep in 1 2 3; do
  for n in 512 1024 2048; do
    f=${1:-400}; [ "$n" = 2048 ] && f=$((f / 4))
    ./gpurun_out/adaptor_demo $n 1.5 $f 2>&1 >/dev/null | grep adaptor_demo
  done
done
