#!/bin/bash
# GPU box (round 5, second call): what a device-to-host copy slows down (sdma_probe2), and what a frame costs along the path
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_second
mkdir -p $O
cd $R
echo "== sdma probe 2"; timeout 300 $R/_variants/sdma_probe2 > $O/sdma_probe2.txt 2>&1; cat $O/sdma_probe2.txt
P="0,0.05,0.125,0.25,0.375,0.5,0.625,0.75,0.875,0.95,1"
echo "== along the path (shipped)"; IDENTICAL=0 PATHS= POSITIONS=$P REPS=40 timeout 900 python tools/ahead_time.py > $O/along_path.txt 2>&1; grep -E "camera at" $O/along_path.txt
echo "== along the path (tile counters)"; KBE_LIB_PATH=$R/_variants/stats.so IDENTICAL=0 PATHS= POSITIONS=$P REPS=8 timeout 900 python tools/ahead_time.py > $O/along_path_stats.txt 2>&1; grep -E "camera at" $O/along_path_stats.txt
