#!/bin/bash
# quick check of a kernel change: LK / disparity parity tests, then the default bench without the CPU and upload legs
set -o pipefail
mkdir -p gpurun_out/r03
tag=${1:-ab}
timeout -k 10 600 python -u -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_sweep.py tests/test_gpu_dense.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03/${tag}_tests.log 2>&1 || { tail -30 gpurun_out/r03/${tag}_tests.log; exit 1; }
tail -2 gpurun_out/r03/${tag}_tests.log
timeout -k 10 300 python bench.py --no-cpu --no-upload-leg --no-latency > gpurun_out/r03/${tag}_bench.json 2> gpurun_out/r03/${tag}_bench.err || { tail -20 gpurun_out/r03/${tag}_bench.err; exit 1; }
python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/r03/${tag}_bench.json') if l.startswith('{')][-1])
print('${tag}', round(j['value'],1), 'f/s', {k:v for k,v in j['config']['stage_ms_one_batch_in_flight'].items() if v>0.1})
PY
