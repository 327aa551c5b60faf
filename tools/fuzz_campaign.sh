# developer tool: every fuzzer for a while on the current build; one summary line each -> gpurun_out/fuzz_campaign.txt
#   bash tools/fuzz_campaign.sh [seed] [scale]     (scale 1 = about 18 minutes)
cd $GRAFT_REPO_ROOT
S=${1:-4242}; X=${2:-1}
O=gpurun_out/fuzz_campaign.txt; : > $O
run() { name=$1; secs=$2; seed=$3; timeout -k 10 $((secs + 120)) python tools/$name.py $secs $seed > gpurun_out/$name.log 2>&1; echo "$name seed $seed ${secs}s: $(tail -1 gpurun_out/$name.log)" | tee -a $O; }
run fuzz_gpu $((300 * X)) $S
run fuzz_big $((200 * X)) $((S + 1))
run fuzz_tiny $((150 * X)) $((S + 2))
run fuzz_slab $((120 * X)) $((S + 3))
run fuzz_stream $((100 * X)) $((S + 4))
run fuzz_api $((100 * X)) $((S + 5))
run fuzz_pointcloud $((100 * X)) $((S + 6))
