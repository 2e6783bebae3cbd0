#!/bin/bash
# GPU box: the round's profile artefacts, written under gpurun_out/$1 (copy what is to be judged into profiles/).
#   gpurun --timeout 2400 -- 'bash tools/gpu_profile.sh r03'
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the default bench command (frames delivered to pinned host memory)
timeout 900 python $R/bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json
# 2. the same default command under rocprofv3 --kernel-trace --stats
rm -rf /tmp/ks
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ks -o b --output-format csv -- python $R/bench.py --no-cpu-baseline 2> $OUT/rocprof.err | tail -1 > $OUT/bench_under_rocprof.json
cp /tmp/ks/b_kernel_stats.csv $OUT/bench_kernel_stats.csv
python $R/tools/kernel_times.py /tmp/ks/b_kernel_trace.csv > $OUT/bench_kernel_medians.txt
# 2b. the same with one lane: kernels of different frames never overlap, so the per-kernel averages are those of isolated launches
rm -rf /tmp/ks1
KBE_LANES=1 KBE_HOST_LANES=1 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ks1 -o b --output-format csv -- python $R/bench.py --no-cpu-baseline 2>> $OUT/rocprof.err | tail -1 > $OUT/bench_lanes1_under_rocprof.json
cp /tmp/ks1/b_kernel_stats.csv $OUT/bench_lanes1_kernel_stats.csv
python $R/tools/kernel_times.py /tmp/ks1/b_kernel_trace.csv > $OUT/bench_lanes1_kernel_medians.txt
# 2c. the other scatter route (KBE_FUSED=0: k_project + k_tiles, buckets of records and the z-buffer in HBM), one lane
rm -rf /tmp/ks2
KBE_FUSED=0 KBE_LANES=1 KBE_HOST_LANES=1 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ks2 -o b --output-format csv -- python $R/bench.py --no-cpu-baseline 2>> $OUT/rocprof.err | tail -1 > $OUT/bench_bucket_lanes1_under_rocprof.json
cp /tmp/ks2/b_kernel_stats.csv $OUT/bench_bucket_lanes1_kernel_stats.csv
KBE_FUSED=0 timeout 900 python $R/bench.py --no-cpu-baseline 2>> $OUT/bench.err | tail -1 > $OUT/bench_bucket.json
# 2c'. short videos (the driver's --steps 20 / 75)
for k in 20 75; do timeout 600 python $R/bench.py --no-cpu-baseline --steps $k --warmup 5 2>> $OUT/bench.err | tail -1 > $OUT/bench_steps$k.json; done
# 2c''. the other BASELINE configurations, each a bench line with the `roofline` of its dominant kernel: the dolly
# zoom (configs[3] 4a: its fill), 2048^2 from 16.8 M points (configs[4]), 512^2 (configs[1]'s frame loop)
timeout 900 python $R/bench.py --no-cpu-baseline --dolly --steps 256 --warmup 32 2>> $OUT/bench.err | tail -1 > $OUT/bench_dolly.json
timeout 900 python $R/bench.py --no-cpu-baseline --size 2048 --upsample 2 --steps 64 --warmup 8 2>> $OUT/bench.err | tail -1 > $OUT/bench_config4.json
timeout 900 python $R/bench.py --no-cpu-baseline --size 512 --steps 1024 --warmup 64 2>> $OUT/bench.err | tail -1 > $OUT/bench_512.json
# 2c'''. BASELINE configs[1] as a whole and the 1024^2 partial-convolution forward (bench.py --pipeline), twice: the first process on a fresh
# box runs MIOpen's find step once (Pipeline(miopen_find='auto')), the second runs in immediate mode on the tuned find-db
for k in first second; do timeout 900 python $R/bench.py --pipeline --steps 10 --warmup 2 2>> $OUT/bench.err | tail -1 > $OUT/pipeline_${k}_process.json; done
# 2d. multi-rank code path on this one GPU (gloo; ranks share the device: a functional check, not a measurement)
KBE_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 $R/bench.py --gpus 2 --steps 32 --warmup 4 2>> $OUT/bench.err | tail -1 > $OUT/bench_2ranks_gloo_one_gpu.json
# 2d'. the same code path with configs[4]'s cloud: a 470 MB broadcast (gloo here: a functional check of the size, not an xGMI figure)
KBE_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 $R/bench.py --gpus 2 --size 2048 --upsample 2 --steps 8 --warmup 2 2>> $OUT/bench.err | tail -1 > $OUT/bench_config4_2ranks_gloo_one_gpu.json
# 3. HBM traffic: one PMC pass per counter (no trace domains alongside)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c
  timeout 900 rocprofv3 --pmc $c -d /tmp/pm_$c -o c --output-format csv -- python $R/tools/pmc_traffic.py > $OUT/pmc_$c.log 2>&1
done
python $R/tools/pmc_report.py /tmp/pm_FETCH_SIZE/c_counter_collection.csv /tmp/pm_WRITE_SIZE/c_counter_collection.csv > $OUT/hbm_traffic.json
# 3b. wave-level instruction counts of the scatter kernels, one frame per launch, both routes (the VALU-issue roofline of bench.py)
for fused in 1 0; do
  rm -rf /tmp/pi_$fused
  KBE_FUSED=$fused KBE_LANES=1 KBE_FILL_GROUP=1 FRAMES=17 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d /tmp/pi_$fused -o c --output-format csv -- python $R/tools/frame_once.py > $OUT/pmc_insts_$fused.log 2>&1
done
python $R/tools/pmc_insts.py /tmp/pi_1/c_counter_collection.csv /tmp/pi_0/c_counter_collection.csv > $OUT/scatter_insts.json
# 3b'. the scatter's GROUP launches under the same counters (tools/ahead_time.py: 1, 2, 4, 8, 12 frames per launch alone on a stream): what
# bench.py prices the launch of the timed region with -- the frames of a group share their candidate lists, which the one-frame launch cannot show
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" FETCH_SIZE WRITE_SIZE; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/pg_$tag
  REPS=6 timeout 600 rocprofv3 --pmc $c -d /tmp/pg_$tag -o c --output-format csv -- python $R/tools/ahead_time.py > $OUT/pmc_group_$tag.log 2>&1
  python $R/tools/pmc_by_grid.py /tmp/pg_$tag/c_counter_collection.csv k_frame_group_ahead k_frame_group --json > $OUT/pmc_group_$tag.json
done
python $R/tools/pmc_group_report.py $OUT/pmc_group_SQ_INSTS_VALU.json $OUT/pmc_group_FETCH_SIZE.json $OUT/pmc_group_WRITE_SIZE.json $OUT/hbm_traffic.json $OUT/scatter_insts.json > $OUT/scatter_group_counters.txt
# 3c. configs[4] on one lane: HBM bytes per launch of its scatter (FETCH_SIZE in 2 KB units, WRITE_SIZE in 1 KB units: tools/pmc_report.py's calibration)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pd_$c
  SIZE=2048 UPSAMPLE=2 CLOUD=raw KBE_LANES=1 FRAMES=16 REPS=1 timeout 400 rocprofv3 --pmc $c -d /tmp/pd_$c -o c --output-format csv -- python $R/tools/throughput.py > /tmp/pd.log 2>&1 || tail -3 /tmp/pd.log
  python $R/tools/pmc_by_grid.py /tmp/pd_$c/c_counter_collection.csv k_frame_group_ahead k_frame_group k_place k_project k_tiles 2>&1 | cut -c1-300
done > $OUT/config4_traffic.txt
python $R/tools/pmc_config4_report.py $OUT/config4_traffic.txt $OUT/hbm_traffic.json > $OUT/hbm_traffic_config4.json
# 4. other workloads (device-only and delivered), both routes where it matters
(
for env in "SIZE=512" "CLOUD=raw" "DOLLY=1" "SIZE=2048 CLOUD=raw" "SIZE=2048 UPSAMPLE=2 CLOUD=raw"; do
  for fused in 0 1; do
    echo "== $env KBE_FUSED=$fused"
    env $env KBE_FUSED=$fused FRAMES=128 REPS=3 timeout 300 python $R/tools/throughput.py 2>/dev/null | tail -1
  done
done
) > $OUT/other_workloads.txt
# 4b. both routes' scatter launches with 1..4 frames each, alone on a stream: rocprofv3 kernel trace of exactly those, by frames per launch
rm -rf /tmp/kg
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kg -o b --output-format csv -- python $R/tools/scatter_time.py > $OUT/scatter_group.txt 2>/dev/null
if [ -f /tmp/kg/b_kernel_stats.csv ]; then cp /tmp/kg/b_kernel_stats.csv $OUT/scatter_group_kernel_stats.csv; python $R/tools/kernel_times_by_grid.py /tmp/kg/b_kernel_trace.csv k_place k_frame k_project k_tiles > $OUT/scatter_group_by_frames_per_launch.txt; fi
# 4c. the same for the one-launch scatter (k_frame_group_ahead: the tiles of a group + the placements of the next), against the two launches
rm -rf /tmp/ka
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ka -o b --output-format csv -- python $R/tools/ahead_time.py > $OUT/scatter_ahead.txt 2>/dev/null
if [ -f /tmp/ka/b_kernel_stats.csv ]; then cp /tmp/ka/b_kernel_stats.csv $OUT/scatter_ahead_kernel_stats.csv; python $R/tools/kernel_times_by_grid.py /tmp/ka/b_kernel_trace.csv k_place k_frame > $OUT/scatter_ahead_by_frames_per_launch.txt; fi
# 5. the dolly zoom (frames with very many holes: the distance-table fill): per-kernel stats of its frame loop, and what the fill does
rm -rf /tmp/kd
DOLLY=1 REPS=2 FRAMES=128 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kd -o b --output-format csv -- python $R/tools/throughput.py > $OUT/dolly_under_rocprof.txt 2>/dev/null
if [ -f /tmp/kd/b_kernel_stats.csv ]; then cp /tmp/kd/b_kernel_stats.csv $OUT/dolly_kernel_stats.csv; fi
timeout 600 python $R/tools/fill_stats.py 2>/dev/null | grep "^holes" > $OUT/dolly_fill_stats.txt
ls -la $OUT; cat $OUT/bench.json; grep -E 'k_tiles|k_project|k_place|k_frame|k_fill_holes|k_crop|copy|Copy' $OUT/bench_kernel_stats.csv $OUT/bench_lanes1_kernel_stats.csv $OUT/bench_bucket_lanes1_kernel_stats.csv | cut -c1-220; cat $OUT/hbm_traffic.json; cat $OUT/other_workloads.txt
