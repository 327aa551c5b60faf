# developer tool: SQ counters + kernel stats of the bench step: bash tools/prof_sq.sh <tag>  ->  gpurun_out/sq_<tag>.json, gpurun_out/ks_<tag>.csv
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/sqtmp_${1:-x}
rm -rf $O; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 5 --warmup 2 > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $O/sq1 -- $B --steps 2 --warmup 1 > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq2 -- $B --steps 2 --warmup 1 > $O/sq2.log 2>&1
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/ks_${1:-x}.csv
python tools/pmc_summary.py $O/sq1 $O/sq2 > gpurun_out/sq_${1:-x}.json
rm -rf $O
