#!/bin/bash
# developer tool: LDS staging capacity of the R=2 sweep (k+1 > 64) vs cell-occupancy factor
for cap in 896 768 640; do
  export PCT_EXTRA_FLAGS="-DPCT_STAGE_CAP2=$cap"
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  echo "== CAP2=$cap"
  for k in 80 100; do timeout -k 5 120 python tools/tune_factor.py 1000000 $k 0.35 0.4 0.45 0.5 | cut -c1-80; done
done
