# A/B of build flags on one box: tools/ab.sh "<flags A>" "<flags B>" ...   (each variant is compiled on the box)
for v in "$@"; do
  echo "[$v]"
  PCT_EXTRA_FLAGS="$v" timeout -k 10 400 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['stage_ms'].items()})"
done
