# the staggered start of a 2048^2 frame on / off: developer build (make -C watersurfacerendering_amd/csrc variant NAME=dev DEFS=-DOCEAN_DEVELOPER),
# OCEAN_RAMP_Z / _B / _D = ramps of the three launches in 10 ns (0 = off); frame hashes, per-kernel times, the synchronous call, pipelined bursts
export OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_dev.so
off() { export OCEAN_RAMP_Z=0 OCEAN_RAMP_B=0 OCEAN_RAMP_D=0; }
on() { unset OCEAN_RAMP_Z OCEAN_RAMP_B OCEAN_RAMP_D; }
for cfg in "2048 1" "2048 1 3"; do on; echo "[ramp] $(python3 tools/frame_hash.py $cfg | sed 's/zpass grid.*sha/sha/')"; off; echo "[none] $(python3 tools/frame_hash.py $cfg | sed 's/zpass grid.*sha/sha/')"; done
for rep in 1 2 3; do
  for m in on off; do $m
    echo "[$m] $(python3 tools/kernel_times.py 2048 1 | cut -c1-110) | $(python3 tools/sync_cost.py 2048 2>&1 | tail -1 | cut -c1-80)"
    echo "[$m] $(python3 tools/burst_probe.py 3 2>&1 | grep 'K=20:\|K=100:\|K=1000:' | sed 's/enqueue.*-> //' | sed 's/ us.frame; best/ best/' | tr '\n' ' ')"
  done
done
