for iq in 8 10 12 16 24; do
  echo "items_q=$iq"; PCT_ITEMS_Q=$iq timeout -k 10 200 python tools/tune_factor.py 1000000 50 0.50 0.55 0.60 | cut -c1-75
done
