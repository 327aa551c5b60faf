# developer tool: FETCH_SIZE of the sweep kernel with and without the XCD-aware item order
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1"
rm -rf gpurun_out/fab; mkdir -p gpurun_out/fab
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/fab/on -- $B > gpurun_out/fab/on.log 2>&1
PCT_NO_XCD_MAP=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/fab/off -- $B > gpurun_out/fab/off.log 2>&1
for v in on off; do echo $v; python tools/pmc_summary.py gpurun_out/fab/$v | python -c "import json,sys; d=json.load(sys.stdin); print({k: round(v['FETCH_SIZE']/1024,1) for k,v in d.items() if 'FETCH_SIZE' in v})"; done
rm -rf gpurun_out/fab/on gpurun_out/fab/off
