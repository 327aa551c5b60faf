#!/bin/bash
# developer tool: rebuild with different LDS staging capacities and time the sweep
for cap in 448 512 576; do
  export PCT_EXTRA_FLAGS="-DPCT_STAGE_CAP=$cap"
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  echo "== CAP=$cap"
  timeout -k 5 120 python tools/tune_factor.py 1000000 50 0.45 0.5 0.55 0.6 | cut -c1-80
  timeout -k 5 120 python tools/tune_factor.py 1000000 30 0.5 0.6 0.7 | cut -c1-80
done
