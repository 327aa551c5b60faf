# A/B of build flags on one box: tools/ab.sh "<flags A>" "<flags B>" ...
for v in "$@"; do
  echo "[$v]"; PCT_EXTRA_FLAGS="$v" timeout -k 10 400 python tools/tune_factor.py 1000000 50 0.55 0.55 | cut -c1-72
done
