# usage: ab_env.sh VAR "<values>" N tiles depths   (developer A/B over an env switch, 3 interleaved repeats)
for rep in 1 2 3; do
  for V in $2; do
    if [ "$V" = "unset" ]; then unset $1; else export $1=$V; fi
    echo -n "[$1=$V] "; python tools/depth_batch.py $3 $4 $5
  done
done
