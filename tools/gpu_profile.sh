#!/bin/bash
# GPU box: the round's profile artefacts, written under gpurun_out/$1 (copy what is to be judged into profiles/).
#   gpurun --timeout 1500 -- 'bash tools/gpu_profile.sh r01'
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the default bench command, plain, and with the PCIe-inclusive leg
timeout 900 python $R/bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json
timeout 900 python $R/bench.py --host-delivery --no-cpu-baseline 2>> $OUT/bench.err | tail -1 > $OUT/bench_host_delivery.json
# 2. the same default command under rocprofv3 --kernel-trace --stats
rm -rf /tmp/ks
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ks -o b --output-format csv -- python $R/bench.py --no-cpu-baseline 2> $OUT/rocprof.err | tail -1 > $OUT/bench_under_rocprof.json
cp /tmp/ks/b_kernel_stats.csv $OUT/bench_kernel_stats.csv
python $R/tools/kernel_times.py /tmp/ks/b_kernel_trace.csv > $OUT/bench_kernel_medians.txt
# 2b. the same with one lane (KBE_LANES=1): kernels never overlap, so the per-kernel averages are those of isolated launches
rm -rf /tmp/ks1
KBE_LANES=1 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ks1 -o b --output-format csv -- python $R/bench.py --no-cpu-baseline 2>> $OUT/rocprof.err | tail -1 > $OUT/bench_lanes1_under_rocprof.json
cp /tmp/ks1/b_kernel_stats.csv $OUT/bench_lanes1_kernel_stats.csv
python $R/tools/kernel_times.py /tmp/ks1/b_kernel_trace.csv > $OUT/bench_lanes1_kernel_medians.txt
# 2c. multi-rank code path on this one GPU (gloo; ranks share the device: a functional check, not a measurement)
KBE_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 $R/bench.py --gpus 2 --steps 32 --warmup 4 2>> $OUT/bench.err | tail -1 > $OUT/bench_2ranks_gloo_one_gpu.json
# 3. HBM traffic: one PMC pass per counter (no trace domains alongside)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c
  timeout 900 rocprofv3 --pmc $c -d /tmp/pm_$c -o c --output-format csv -- python $R/tools/pmc_traffic.py > $OUT/pmc_$c.log 2>&1
done
python $R/tools/pmc_report.py /tmp/pm_FETCH_SIZE/c_counter_collection.csv /tmp/pm_WRITE_SIZE/c_counter_collection.csv > $OUT/hbm_traffic.json
# 4. a clean trace of 33 frames (no bench differencing, nothing else on the device): per-kernel medians
rm -rf /tmp/o
KBE_LANES=1 FRAMES=33 timeout 600 rocprofv3 --kernel-trace -d /tmp/o -o t --output-format csv -- python $R/tools/frame_once.py > /dev/null 2>&1
python $R/tools/kernel_times.py /tmp/o/t_kernel_trace.csv > $OUT/frame_loop_kernel_medians.txt
ls -la $OUT; cat $OUT/bench.json; cat $OUT/frame_loop_kernel_medians.txt; grep -E 'k_tiles|k_project|k_fill_holes|k_crop' $OUT/bench_kernel_stats.csv $OUT/bench_lanes1_kernel_stats.csv | cut -c1-200; cat $OUT/bench_2ranks_gloo_one_gpu.json | cut -c1-300
