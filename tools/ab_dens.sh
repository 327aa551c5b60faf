# A/B of build flags on the density probe: tools/ab_dens.sh "<flags A>" "<flags B>" ...
for v in "$@"; do
  echo "[$v]"
  PCT_EXTRA_FLAGS="$v" timeout -k 10 400 python tools/density_probe.py 1000000 2>&1 | grep LEVELS
done
