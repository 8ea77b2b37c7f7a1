#!/bin/bash
# 4K x 256 disparities: bench line, kernel stats and counter passes into gpurun_out/prof4k/ (run from the repository root on a GPU box)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof4k
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="$root/bench.py"
A="--workload 4k --no-cpu --no-upload-leg --no-latency --no-collective --frame-cache /tmp/vo_frames"
timeout -k 10 400 python3 $B --workload 4k --steps 50 --frame-cache /tmp/vo_frames > $out/bench_4k.json 2> $out/bench_4k.err
timeout -k 10 200 python3 $B $A --inflight 1 --steps 1 --warmup 0 > /dev/null 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $out -o inflight1 --output-format csv -- python3 $B $A --steps 3 --warmup 1 --inflight 1 > $out/inflight1.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $out -o fetch --output-format csv -- python3 $B $A --steps 2 --warmup 1 --inflight 1 > $out/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $out -o write --output-format csv -- python3 $B $A --steps 2 --warmup 1 --inflight 1 > $out/write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    -d $out -o sq --output-format csv -- python3 $B $A --steps 2 --warmup 1 --inflight 1 > $out/sq.log 2>&1
ls -la $out
