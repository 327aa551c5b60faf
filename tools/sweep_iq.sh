# developer tool: sweep the queries-per-work-item limit on the bench workload
for iq in 8 10 12 14 16 20 24 32 64; do
  echo -n "items_q=$iq  "
  PCT_ITEMS_Q=$iq timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --warmup 5 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['stage_ms'])"
done
