#!/bin/bash
# GPU box (round 4): configs[4] on the fused route with a variant library: per-kernel durations (one lane) and VALU counters
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
V=$R/_variants/${1:-amu9u8.so}
rm -rf /tmp/kt
KBE_LIB_PATH=$V SIZE=2048 UPSAMPLE=2 CLOUD=raw KBE_FUSED=1 KBE_LANES=1 FRAMES=32 REPS=2 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t --output-format csv -- python $R/tools/throughput.py > /tmp/kt.log 2>&1
tail -1 /tmp/kt.log; head -7 /tmp/kt/t_kernel_stats.csv | cut -c1-150 | tee $O/i_c4_kstats.txt
for set in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU2 SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pc
  KBE_LIB_PATH=$V SIZE=2048 UPSAMPLE=2 CLOUD=raw KBE_FUSED=1 KBE_LANES=1 FRAMES=16 REPS=1 timeout 400 rocprofv3 --pmc $set -d /tmp/pc -o c --output-format csv -- python $R/tools/throughput.py > /tmp/pc.log 2>&1 || tail -3 /tmp/pc.log
  python $R/tools/pmc_by_grid.py /tmp/pc/c_counter_collection.csv k_frame_group_ahead k_frame_group k_place 2>&1 | cut -c1-400
done | tee $O/i_c4_pmc.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c
  KBE_LIB_PATH=$V SIZE=2048 UPSAMPLE=2 CLOUD=raw KBE_FUSED=1 KBE_LANES=1 FRAMES=16 REPS=1 timeout 400 rocprofv3 --pmc $c -d /tmp/pm_$c -o c --output-format csv -- python $R/tools/throughput.py > /tmp/pm.log 2>&1 || tail -3 /tmp/pm.log
  python $R/tools/pmc_by_grid.py /tmp/pm_$c/c_counter_collection.csv k_frame_group_ahead k_frame_group k_place 2>&1 | cut -c1-300
done | tee $O/i_c4_traffic.txt
