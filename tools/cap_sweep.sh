#!/bin/bash
# developer tool: rebuild with different LDS staging capacities and time the sweep
for cap in 512 448 384; do
  export PCT_EXTRA_FLAGS="-DPCT_STAGE_CAP=$cap"
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  echo "== CAP=$cap"
  PCT_STATS=0 timeout -k 5 120 python tools/tune_factor.py 1000000 50 0.4 0.45 0.5 0.55
done
