# developer tool: bench step with prebuilt library variants (tools/build_variant.py): bash tools/ab_lib.sh variants/A.so variants/B.so ...  ("-" = the real library)
for v in "$@"; do
  if [ "$v" = "-" ]; then unset PCT_LIB; else export PCT_LIB=$PWD/$v; fi
  echo -n "[$v] "
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['stage_ms'].items()})"
done
