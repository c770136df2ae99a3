# usage: ab2.sh kernels|frames "<variants>" N tiles [depths]   -- developer A/B, 2 interleaved repeats.
# A variant is LIB or LIB:VAR=VAL[,VAR=VAL...]  (LIB "default" = the shipped library; others = libocean_hip_LIB.so)
what=$1; shift
for rep in 1 2; do
  for V in $1; do
    L=${V%%:*}; E=""; [ "$V" != "$L" ] && E=$(echo "${V#*:}" | tr ',' ' ')
    if [ "$L" = "default" ]; then LIBENV=""; else LIBENV="OCEAN_HIP_LIB=watersurfacerendering_amd/libocean_hip_$L.so"; fi
    echo -n "[$V] "
    if [ "$what" = "kernels" ]; then env $LIBENV $E python tools/kernel_times.py $2 ${3:-1};
    else env $LIBENV $E python tools/depth_batch.py $2 $3 $4; fi
  done
done
