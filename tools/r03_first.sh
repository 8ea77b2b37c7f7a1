#!/bin/bash
# round-3 first GPU pass: new full-size tests, the Python mirror, default bench with the one-rank RCCL gather, dense and 4K lines
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 600 python -u -m pytest tests/test_gpu_fullsize.py tests/test_gpu_voclass_py.py tests/test_gpu_dense.py -x -v -s --timeout 300 --timeout-method thread > gpurun_out/r03/tests_a.log 2>&1 || { tail -30 gpurun_out/r03/tests_a.log; exit 1; }
tail -4 gpurun_out/r03/tests_a.log
timeout -k 10 300 python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err || { tail -30 gpurun_out/r03/bench_default.err; exit 1; }
cut -c1-600 gpurun_out/r03/bench_default.json
timeout -k 10 300 python bench.py --workload dense --steps 40 > gpurun_out/r03/bench_dense.json 2> gpurun_out/r03/bench_dense.err || { tail -30 gpurun_out/r03/bench_dense.err; exit 1; }
cut -c1-600 gpurun_out/r03/bench_dense.json
timeout -k 10 400 python bench.py --workload 4k --steps 50 > gpurun_out/r03/bench_4k.json 2> gpurun_out/r03/bench_4k.err || { tail -30 gpurun_out/r03/bench_4k.err; exit 1; }
cut -c1-600 gpurun_out/r03/bench_4k.json
