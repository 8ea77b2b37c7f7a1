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
B="$root/bench.py"
C="--frame-cache /tmp/vo_frames"   # rendered once by the first (unprofiled) run; profiled runs must not fork workers
timeout -k 10 200 python3 $B $C > $out/bench_default.json 2> $out/bench_default.err
timeout -k 10 100 python3 $B $C --inflight 1 --steps 1 --warmup 0 --no-cpu --no-upload-leg > /dev/null 2>&1   # fills the frame cache of the one-batch-in-flight passes
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $out -o default --output-format csv -- python3 $B $C --no-cpu --no-upload-leg > $out/default.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $out -o inflight1 --output-format csv -- python3 $B $C --no-cpu --no-upload-leg --steps 10 --warmup 2 --inflight 1 > $out/inflight1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out -o fetch --output-format csv -- python3 $B $C --steps 3 --warmup 1 --no-cpu --no-upload-leg --inflight 1 > $out/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $out -o write --output-format csv -- python3 $B $C --steps 3 --warmup 1 --no-cpu --no-upload-leg --inflight 1 > $out/write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    -d $out -o sq --output-format csv -- python3 $B $C --steps 3 --warmup 1 --no-cpu --no-upload-leg --inflight 1 > $out/sq.log 2>&1
if [ -x $root/gpurun_variants/valu_rate ]; then timeout -k 10 60 $root/gpurun_variants/valu_rate > $out/micro_valu_rate.txt; fi
if [ -x $root/gpurun_variants/hbm_rate ]; then timeout -k 10 60 $root/gpurun_variants/hbm_rate > $out/micro_hbm_rate.txt; fi
ls -la $out
