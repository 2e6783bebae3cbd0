#!/bin/bash
# GPU box (round 4): step 3b' of tools/gpu_profile.sh on its own -- the scatter's group launches under the counters, merged into copies
# of the committed profiles/r04_hbm_traffic.json and r04_scatter_insts.json (written to gpurun_out/$1)
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r04h}
mkdir -p $OUT
cp $R/profiles/r04_hbm_traffic.json $OUT/hbm_traffic.json
cp $R/profiles/r04_scatter_insts.json $OUT/scatter_insts.json
cd /tmp && export TMPDIR=/tmp
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" FETCH_SIZE WRITE_SIZE; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/pg_$tag
  REPS=6 timeout 600 rocprofv3 --pmc $c -d /tmp/pg_$tag -o c --output-format csv -- python $R/tools/ahead_time.py > $OUT/pmc_group_$tag.log 2>&1
  python $R/tools/pmc_by_grid.py /tmp/pg_$tag/c_counter_collection.csv k_frame_group_ahead k_frame_group --json > $OUT/pmc_group_$tag.json
done
python $R/tools/pmc_group_report.py $OUT/pmc_group_SQ_INSTS_VALU.json $OUT/pmc_group_FETCH_SIZE.json $OUT/pmc_group_WRITE_SIZE.json $OUT/hbm_traffic.json $OUT/scatter_insts.json | tee $OUT/scatter_group_counters.txt
