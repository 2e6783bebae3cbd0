#!/bin/bash
# GPU box (round 4): the phase timeline of the scatter's waves (tools/frame_probe.py) and the instruction-class PMC passes over the
# scatter launches and over the calibration streams.  Output: gpurun_out/r04/.
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FRAMES=8 timeout 600 python $R/tools/frame_probe.py > $O/frame_probe_8.txt 2>&1; cat $O/frame_probe_8.txt | tail -14
FRAMES=1 timeout 600 python $R/tools/frame_probe.py > $O/frame_probe_1.txt 2>&1; cat $O/frame_probe_1.txt | tail -14
hipcc --offload-arch=gfx950 -O2 $R/tools/valu_rate.hip -o /tmp/valu_rate
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
           "SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  rm -rf /tmp/pc$i /tmp/pd$i
  timeout 600 rocprofv3 --pmc $set -d /tmp/pc$i -o c --output-format csv -- /tmp/valu_rate 8 > /tmp/pc$i.log 2>&1 || tail -5 /tmp/pc$i.log
  python $R/tools/pmc_by_grid.py /tmp/pc$i/c_counter_collection.csv > $O/valu_rate_classes$i.txt
  REPS=8 timeout 600 rocprofv3 --pmc $set -d /tmp/pd$i -o c --output-format csv -- python $R/tools/ahead_time.py > /tmp/pd$i.log 2>&1 || tail -5 /tmp/pd$i.log
  python $R/tools/pmc_by_grid.py /tmp/pd$i/c_counter_collection.csv --json k_frame_group_ahead k_frame_group k_place k_frame_ahead > $O/pmc_classes$i.json
  i=$((i+1))
done
python - <<P
import json
for i in range(2):
    d = json.load(open('$O/pmc_classes%d.json' % i))
    for k, v in d.items():
        if 'k_frame_group_ahead grid=4194304' in k: print(k, v)
P
