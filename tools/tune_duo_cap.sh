# developer tool: k_knn_duo with 768 (default) against 1024 staged slots (python tools/build_variant.py variants/duo1024.so pct_knn.hip=-DPCT_DUO_CAP=1024)
# over the occupancy factor at k = 80 / 100 / 127 -- 1024 slots (3 waves per SIMD) lose at k = 80 (0.74 vs 0.62 ms), tie at 100, win at 127 (1.26 vs 2.54)
for lib in "" variants/duo1024.so; do
  if [ -n "$lib" ]; then export PCT_LIB=$PWD/$lib; else unset PCT_LIB; fi
  echo "== lib=${lib:-default}"
  for k in 80 100 127; do echo "k=$k"; PCT_STATS=1 timeout -k 10 100 python tools/tune_factor.py 1000000 $k 0.40 0.45 0.50 0.55 0.62 2>&1 | tail -5 | cut -c1-175; done
done
