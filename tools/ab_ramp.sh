export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_dev.so
for cfg in "2048 1" "2048 1 3"; do unset OCEAN_START_RAMP; echo "[ramp] $(python3 tools/frame_hash.py $cfg | sed 's/zpass grid.*sha/sha/')"; export OCEAN_START_RAMP=0; echo "[none] $(python3 tools/frame_hash.py $cfg | sed 's/zpass grid.*sha/sha/')"; done
for rep in 1 2 3; do
  unset OCEAN_START_RAMP; echo "[ramp] $(python3 tools/kernel_times.py 2048 1 | cut -c1-110) | $(python3 tools/sync_cost.py 2048 2>&1 | tail -1 | cut -c1-80)"
  export OCEAN_START_RAMP=0; echo "[none] $(python3 tools/kernel_times.py 2048 1 | cut -c1-110) | $(python3 tools/sync_cost.py 2048 2>&1 | tail -1 | cut -c1-80)"
done
