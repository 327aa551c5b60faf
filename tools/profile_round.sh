# developer tool: every profile artefact of one build from one box: bash tools/profile_round.sh <tag>  ->  gpurun_out/prof_<tag>/
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_${1:-x}
rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
echo bench done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/stats.log 2>&1
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/write.log 2>&1
echo write done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $O/sq1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/sq1.log 2>&1
echo sq1 done
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/sq2.log 2>&1
echo sq2 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lat -- python3 tools/lattice_probe.py > $O/lattice.log 2>&1
echo lattice done
find $O/lat -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/lattice_kernel_stats.csv
python tools/pmc_summary.py $O/sq1 $O/sq2 > $O/sq.json
python tools/pmc_summary.py $O/fetch $O/write > $O/traffic_raw.json
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
# keep only the summaries (the raw traces are large)
rm -rf $O/stats $O/fetch $O/write $O/sq1 $O/sq2 $O/lat
ls -la $O
