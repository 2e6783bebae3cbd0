#!/bin/bash
# GPU box: round 6's profile artefacts, written under gpurun_out/$1 (copy what is to be judged into profiles/).  As tools/gpu_profile.sh,
# with the scatter's launch traced and counted on CONSECUTIVE cameras of the product's 75-step path.
#   gpurun --timeout 3000 -- 'bash tools/batches/gpu_r06_profile.sh r06p'
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r06p}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the driver's command, the default command, the product's length
timeout 900 python $R/bench.py --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/bench_steps20.json
timeout 900 python $R/bench.py 2>> $OUT/bench.err | tail -1 > $OUT/bench.json
timeout 900 python $R/bench.py --no-cpu-baseline --steps 75 --warmup 20 2>> $OUT/bench.err | tail -1 > $OUT/bench_steps75.json
timeout 900 python $R/bench.py --no-cpu-baseline --video-frames 128 --warmup 20 2>> $OUT/bench.err | tail -1 > $OUT/bench_video128_strong.json
# 2. the driver's command under rocprofv3 --kernel-trace --stats
rm -rf /tmp/ks
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ks -o b --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 2> $OUT/rocprof.err | tail -1 > $OUT/bench_steps20_under_rocprof.json
cp /tmp/ks/b_kernel_stats.csv $OUT/bench_steps20_kernel_stats.csv
python $R/tools/kernel_times.py /tmp/ks/b_kernel_trace.csv > $OUT/bench_steps20_kernel_medians.txt
# 3. the scatter launch ALONE on a stream under the kernel trace: (a) twelve consecutive cameras of the 75-step path per launch, (b) of the 20- and
# the 1024-step path, (c) twelve copies of one camera (round 4's shape)
for spec in "consecutive75:IDENTICAL=0 PATHS=75" "consecutive20:IDENTICAL=0 PATHS=20" "consecutive1024:IDENTICAL=0 PATHS=1024" "identical:IDENTICAL=12 PATHS="; do
  tag=${spec%%:*}; envs=${spec#*:}
  rm -rf /tmp/ka
  env $envs SKIP_CHECK=1 LAUNCH_FRAMES=12 REPS=60 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ka -o b --output-format csv -- python $R/tools/ahead_time.py > $OUT/scatter_ahead_$tag.txt 2>/dev/null
  if [ -f /tmp/ka/b_kernel_stats.csv ]; then cp /tmp/ka/b_kernel_stats.csv $OUT/scatter_ahead_${tag}_kernel_stats.csv; python $R/tools/kernel_times_by_grid.py /tmp/ka/b_kernel_trace.csv k_place k_frame > $OUT/scatter_ahead_${tag}_by_frames_per_launch.txt; fi
done
# 4. HBM traffic and instruction counts: one PMC pass per counter set (no trace domains alongside); one frame per launch first (the calibration
# kernels ride in tools/pmc_traffic.py), then the GROUP launch on consecutive cameras of the 75-step path
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c
  timeout 900 rocprofv3 --pmc $c -d /tmp/pm_$c -o c --output-format csv -- python $R/tools/pmc_traffic.py > $OUT/pmc_$c.log 2>&1
done
python $R/tools/pmc_report.py /tmp/pm_FETCH_SIZE/c_counter_collection.csv /tmp/pm_WRITE_SIZE/c_counter_collection.csv > $OUT/hbm_traffic.json
for fused in 1 0; do
  rm -rf /tmp/pi_$fused
  KBE_FUSED=$fused KBE_LANES=1 KBE_FILL_GROUP=1 FRAMES=17 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d /tmp/pi_$fused -o c --output-format csv -- python $R/tools/frame_once.py > $OUT/pmc_insts_$fused.log 2>&1
done
python $R/tools/pmc_insts.py /tmp/pi_1/c_counter_collection.csv /tmp/pi_0/c_counter_collection.csv > $OUT/scatter_insts.json
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" FETCH_SIZE WRITE_SIZE; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/pg_$tag
  IDENTICAL=0 PATHS=75 SKIP_CHECK=1 LAUNCH_FRAMES=12 REPS=14 timeout 600 rocprofv3 --pmc $c -d /tmp/pg_$tag -o c --output-format csv -- python $R/tools/ahead_time.py > $OUT/pmc_group_$tag.log 2>&1
  python $R/tools/pmc_by_grid.py /tmp/pg_$tag/c_counter_collection.csv k_frame_group_ahead k_frame_group --json > $OUT/pmc_group_$tag.json
done
python $R/tools/pmc_group_report.py $OUT/pmc_group_SQ_INSTS_VALU.json $OUT/pmc_group_FETCH_SIZE.json $OUT/pmc_group_WRITE_SIZE.json $OUT/hbm_traffic.json $OUT/scatter_insts.json > $OUT/scatter_group_counters.txt
# (the same three passes on twelve copies of one camera, for the comparison with round 4's figures)
cp $OUT/hbm_traffic.json $OUT/hbm_traffic_identical.json; cp $OUT/scatter_insts.json $OUT/scatter_insts_identical.json
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" FETCH_SIZE WRITE_SIZE; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/ph_$tag
  IDENTICAL=12 PATHS= SKIP_CHECK=1 REPS=14 timeout 600 rocprofv3 --pmc $c -d /tmp/ph_$tag -o c --output-format csv -- python $R/tools/ahead_time.py > $OUT/pmc_identical_$tag.log 2>&1
  python $R/tools/pmc_by_grid.py /tmp/ph_$tag/c_counter_collection.csv k_frame_group_ahead k_frame_group --json > $OUT/pmc_identical_$tag.json
done
python $R/tools/pmc_group_report.py $OUT/pmc_identical_SQ_INSTS_VALU.json $OUT/pmc_identical_FETCH_SIZE.json $OUT/pmc_identical_WRITE_SIZE.json $OUT/hbm_traffic_identical.json $OUT/scatter_insts_identical.json > $OUT/scatter_group_counters_identical.txt
# 5. the other BASELINE configurations, each a bench line with the `roofline` of its dominant kernel
timeout 900 python $R/bench.py --no-cpu-baseline --dolly --steps 256 --warmup 32 2>> $OUT/bench.err | tail -1 > $OUT/bench_dolly.json
timeout 900 python $R/bench.py --no-cpu-baseline --size 2048 --upsample 2 --steps 64 --warmup 8 2>> $OUT/bench.err | tail -1 > $OUT/bench_config4.json
timeout 900 python $R/bench.py --no-cpu-baseline --size 512 --steps 1024 --warmup 64 2>> $OUT/bench.err | tail -1 > $OUT/bench_512.json
for k in first second; do timeout 900 python $R/bench.py --pipeline --steps 10 --warmup 2 2>> $OUT/bench.err | tail -1 > $OUT/pipeline_${k}_process.json; done
KBE_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 $R/bench.py --gpus 2 --steps 32 --warmup 4 2>> $OUT/bench.err | tail -1 > $OUT/bench_2ranks_gloo_one_gpu.json
KBE_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 $R/bench.py --gpus 2 --video-frames 128 --warmup 8 2>> $OUT/bench.err | tail -1 > $OUT/bench_video128_2ranks_gloo_one_gpu.json
# 5b. the one command for a multi-GPU node, as a dry run on this box's one GPU (ranks share it, collectives on gloo: no curve)
timeout 900 python $R/tools/scale_report.py --gpus 1,2 --dry-run --out $OUT/scale_report_dry_run.json > $OUT/scale_report_dry_run.txt 2>> $OUT/bench.err
# 5c. configs[4] under the counters (HBM bytes per point of the dense launch; the shape tools/batches/gpu_r05_seventh.sh collected it in)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pd_$c
  SIZE=2048 UPSAMPLE=2 CLOUD=raw KBE_LANES=1 FRAMES=18 REPS=1 timeout 400 rocprofv3 --pmc $c -d /tmp/pd_$c -o c --output-format csv -- python $R/tools/throughput.py > /tmp/pd.log 2>&1 || tail -3 /tmp/pd.log
  python $R/tools/pmc_by_grid.py /tmp/pd_$c/c_counter_collection.csv k_frame_group_ahead k_frame_group k_place 2>&1 | cut -c1-300
done > $OUT/config4_traffic.txt
python $R/tools/pmc_config4_report.py $OUT/config4_traffic.txt $OUT/hbm_traffic.json > $OUT/hbm_traffic_config4.json 2>> $OUT/bench.err
# 6. the GPU suite
cd $R && timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4 > $OUT/gpu_suite.txt
ls -la $OUT; cat $OUT/bench_steps20.json; cat $OUT/scatter_group_counters.txt $OUT/scatter_group_counters_identical.txt; for t in consecutive75 consecutive20 consecutive1024 identical; do echo "== $t"; grep -E "k_frame_group_ahead" $OUT/scatter_ahead_${t}_by_frames_per_launch.txt | head -3; grep -E "consecutive|^12 frame" $OUT/scatter_ahead_$t.txt; done; cat $OUT/gpu_suite.txt
