for v in 1100 6000; do
  PCT_EXTRA_FLAGS="-DPCT_LDS_PAD=$v" timeout -k 10 400 python tools/tune_factor.py 1000000 50 0.55 > gpurun_out/pad_$v.log 2>&1 || exit 1
  echo "PAD $v: $(tail -1 gpurun_out/pad_$v.log)"
done
