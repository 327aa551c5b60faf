# kernel times of one rank's slab-ownership step (developer tool): rank G-1 of G=4 on the 5 M-point egg carton
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_slab -- python tools/shard_probe.py 1250000 random-slab egg 4 > gpurun_out/prof_slab.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_slab/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:18]:
    print(r['Name'][:90], r['Calls'], r['AverageNs'])
PY
