for k in 52 56 60 63; do
  echo "== k=$k pair (default factor)"; timeout -k 10 100 python tools/tune_factor.py 1000000 $k 0 2>&1 | tail -1
  echo "== k=$k duo"; PCT_FAST_R1_MAX=50 timeout -k 10 100 python tools/tune_factor.py 1000000 $k 0.40 0.46 0.52 0.60 2>&1 | tail -4
done
for k in 64 80 100 127; do
  echo "== k=$k duo"; PCT_STATS=1 timeout -k 10 100 python tools/tune_factor.py 1000000 $k 0.30 0.36 0.42 0.47 0.52 0.58 0.65 2>&1 | tail -7
done
