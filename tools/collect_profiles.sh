#!/bin/bash
# Collects the measurements kept under profiles/ on a GPU box, into gpurun_out/prof/ :
#   tools/collect_profiles.sh            (run from the repository root; every pass is its own short process)
# Afterwards, on any machine:
#   tools/pmc_traffic.py kitti gpurun_out/prof/fetch_counter_collection.csv gpurun_out/prof/write_counter_collection.csv
#   tools/pmc_sq.py gpurun_out/prof/sq_counter_collection.csv profiles/rNN_sq_summary.json
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# bench.py asks for 8 hardware queues itself, but under the profiler the runtime is up before main(): export it here
export GPU_MAX_HW_QUEUES=8
B="$root/bench.py"
C="--frame-cache /tmp/vo_frames"   # rendered once by the first (unprofiled) run; profiled runs must not fork workers
# profiled runs: one batch size only (no one-pair latency pass), no communicator, no CPU or upload legs
P="--no-cpu --no-upload-leg --no-latency --no-collective"
timeout -k 10 300 python3 $B $C > $out/bench_default.json 2> $out/bench_default.err
# the short run the round-end driver makes (its fill, drain and result gather weigh 5-7 % at 20 steps)
timeout -k 10 300 python3 $B $C --steps 20 --warmup 5 > $out/bench_steps20.json 2> $out/bench_steps20.err
timeout -k 10 100 python3 $B $C --inflight 1 --steps 1 --warmup 0 $P > /dev/null 2>&1   # fills the frame cache of the one-batch-in-flight passes
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $out -o default --output-format csv -- python3 $B $C $P > $out/default.log 2>&1
# one batch in flight AND one stream per context: a batch's tracking chain otherwise runs beside its disparity chain, and a
# kernel's duration would include the kernels it shares the GPU with (bench.py's own kernel-duration pass does the same
# through vo_set_timing(2))
export VO_SINGLE_STREAM=1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $out -o inflight1 --output-format csv -- python3 $B $C $P --steps 10 --warmup 2 --inflight 1 > $out/inflight1.log 2>&1
# the counter passes stay at 42 pairs per launch, the batch profiles/traffic.json and valu.json record (bench.py scales them)
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out -o fetch --output-format csv -- python3 $B $C --batch 42 --steps 3 --warmup 1 $P --inflight 1 > $out/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $out -o write --output-format csv -- python3 $B $C --batch 42 --steps 3 --warmup 1 $P --inflight 1 > $out/write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    -d $out -o sq --output-format csv -- python3 $B $C --batch 42 --steps 3 --warmup 1 $P --inflight 1 > $out/sq.log 2>&1
unset VO_SINGLE_STREAM GPU_MAX_HW_QUEUES   # the dense-tracking runs keep the default 4 queues
# the dense-tracking configuration: its line and its kernel statistics
timeout -k 10 300 python3 $B --workload dense --steps 40 $C > $out/bench_dense.json 2> $out/bench_dense.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out -o dense --output-format csv -- python3 $B --workload dense --steps 6 --warmup 2 --inflight 1 $C $P > $out/dense.log 2>&1
# the 11-sequence workload on this one GPU
export GPU_MAX_HW_QUEUES=8
timeout -k 10 300 python3 $B --workload kitti11 --no-cpu --frames 127 $C > $out/bench_kitti11.json 2> $out/bench_kitti11.err
if [ -x $root/gpurun_variants/valu_rate ]; then timeout -k 10 60 $root/gpurun_variants/valu_rate > $out/micro_valu_rate.txt; fi
if [ -x $root/gpurun_variants/hbm_rate ]; then timeout -k 10 60 $root/gpurun_variants/hbm_rate > $out/micro_hbm_rate.txt; fi
ls -la $out
