#!/bin/bash
# GPU box (round 4): the chip's VALU issue rate per kind of instruction (tools/valu_rate.hip), what the SQ
# counters read on those known instruction streams, the scatter launches alone on a stream (tools/ahead_time.py) and the busy /
# wait PMC passes over the same launches.  Everything lands in gpurun_out/r04/.
#   gpurun --timeout 1500 -- 'bash tools/batches/gpu_r04_calibrate.sh'
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 $R/tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate > $O/valu_rate.txt 2>&1
tail -3 $O/valu_rate.txt
# the counters on the calibration streams (five waves per SIMD, the scatter's occupancy)
rm -rf /tmp/pv
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pv -o c --output-format csv -- /tmp/valu_rate 5 > /tmp/pv.log 2>&1 || tail -5 /tmp/pv.log
python $R/tools/pmc_by_grid.py /tmp/pv/c_counter_collection.csv > $O/valu_rate_pmc.txt
tail -3 $O/valu_rate_pmc.txt
[ "$SKIP_SCATTER" = 1 ] && exit 0
REPS=40 timeout 600 python $R/tools/ahead_time.py > $O/ahead_time.txt 2>&1
tail -5 $O/ahead_time.txt
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES"; do
  rm -rf /tmp/pa$i
  REPS=8 timeout 600 rocprofv3 --pmc $set -d /tmp/pa$i -o c --output-format csv -- python $R/tools/ahead_time.py > /tmp/pa$i.log 2>&1 || tail -5 /tmp/pa$i.log
  python $R/tools/pmc_by_grid.py /tmp/pa$i/c_counter_collection.csv --json k_frame_group_ahead k_frame_group k_place k_frame_ahead > $O/pmc_pass$i.json
  i=$((i+1))
done
python - <<P
import json
for i in range(3):
    d = json.load(open('$O/pmc_pass%d.json' % i))
    for k, v in d.items():
        if 'k_frame_group_ahead' in k: print(k, v)
P
