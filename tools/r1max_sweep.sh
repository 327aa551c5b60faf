for k in 50 56 60 62 63; do
  a=$(PCT_FAST_R1_MAX=64 timeout -k 10 200 python tools/tune_factor.py 1000000 $k 0 | cut -c1-62)
  b=$(PCT_FAST_R1_MAX=48 timeout -k 10 200 python tools/tune_factor.py 1000000 $k 0 | cut -c1-62)
  echo "k=$k R1: $a"; echo "k=$k R2: $b"
done
