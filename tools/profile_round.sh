# developer tool: every profile artefact of one build from one box: bash tools/profile_round.sh <tag>  ->  gpurun_out/prof_<tag>/
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_${1:-x}
rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
echo bench done
B="python3 bench.py --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 5 --warmup 2 > $O/stats.log 2>&1
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --steps 2 --warmup 1 > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --steps 2 --warmup 1 > $O/write.log 2>&1
echo write done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $O/sq1 -- $B --steps 2 --warmup 1 > $O/sq1.log 2>&1
echo sq1 done
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq2 -- $B --steps 2 --warmup 1 > $O/sq2.log 2>&1
echo sq2 done
# the reference's own lattice torus (utils.py:883), same step
PCT_PROBE_NO_STATS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lat -- python3 tools/lattice_probe.py 1000 50 "torus grid" > $O/lattice.log 2>&1
echo lattice done
# the hierarchical cell list on a 1/r^2 scan (what PCT_KNN_AUTO takes for it)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tree -- python3 tools/tree_probe.py 1000000 40 > $O/tree.log 2>&1
echo tree done
find $O/tree -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/tree_kernel_stats.csv
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/lat -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/lattice_kernel_stats.csv
python tools/pmc_summary.py $O/sq1 $O/sq2 > $O/sq.json
python tools/pmc_summary.py $O/fetch $O/write > $O/traffic_raw.json
python - $O <<'PY'
import json, sys
o = sys.argv[1]
d = json.load(open(o + "/traffic_raw.json"))
f, w = d["k_knn_pair"]["FETCH_SIZE"], d["k_knn_pair"]["WRITE_SIZE"]
json.dump({"kernel": "k_knn_pair<false,false> (the fused step's sweep: index table only)",
           "config": "torus 1M seed 1234 k=50, bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras",
           "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
           "correction": "gfx950: FETCH_SIZE reports half of the bytes of 16 B/lane reads (MI355X_MICROARCH.md, HBM) -> doubled; WRITE_SIZE exact",
           "hbm_bytes_per_launch": (2 * f + w) * 1024,
           "other_kernels_KB": {k: v for k, v in d.items() if k != "k_knn_pair"},
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, tools/profile_round.sh)"},
          open(o + "/knn_traffic.json", "w"), indent=1)
PY
# BASELINE configs[3] / [4] on one GPU (C5: k = 80 -> k_knn_duo), each checked against its 2000 sampled reference rows
python bench.py --config c4 --no-extras --no-cpu-baseline --steps 5 --warmup 2 --verify > $O/bench_c4.json 2> $O/bench_c4.err
python bench.py --config c5 --no-extras --no-cpu-baseline --steps 5 --warmup 2 --verify > $O/bench_c5.json 2> $O/bench_c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -- python3 bench.py --config c5 --no-extras --no-cpu-baseline --steps 3 --warmup 1 > $O/c5.log 2>&1
find $O/c5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c5_kernel_stats.csv
rm -rf $O/c5
echo c4 c5 done
# one rank's step of a 4-way job on the unsorted 5 M-point egg carton, ownership by slab (and by index range, for comparison)
python tools/shard_probe.py 1250000 random-slab egg 1,4 > $O/shard_probe_slab.txt 2>&1
python tools/shard_probe.py 1250000 random egg 1,4 >> $O/shard_probe_slab.txt 2>&1
python tools/shard_probe.py 1000000 scan torus > $O/shard_probe_scan.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/slab -- python3 tools/shard_probe.py 1250000 random-slab egg 4 > $O/slab.log 2>&1
find $O/slab -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/slab_rank_step_kernel_stats.csv
rm -rf $O/slab
python tools/redo_probe.py 1000000 50,61,64,80,100 > $O/redo_probe.txt 2>&1
echo shard done
# keep only the summaries (the raw traces are large)
rm -rf $O/stats $O/fetch $O/write $O/sq1 $O/sq2 $O/lat
ls -la $O
