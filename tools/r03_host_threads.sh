#!/bin/bash
# frames/s against the host threads a rank is given (the PnP solvers run on them): default workload and the 11-sequence one
set -o pipefail
mkdir -p gpurun_out/r03
for t in 16 8 4 2; do
  timeout -k 10 300 python bench.py --host-threads $t --no-cpu --no-upload-leg --no-latency --frame-cache gpurun_out/r03/cache > gpurun_out/r03/ht_kitti_$t.json 2> gpurun_out/r03/ht_kitti_$t.err || { tail -20 gpurun_out/r03/ht_kitti_$t.err; exit 1; }
  python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/r03/ht_kitti_$t.json') if l.startswith('{')][-1])
print('kitti host_threads=$t', round(j['value'],1), 'pnp_host ms', j['config']['stage_ms_timed_region'].get('pnp_host'))
PY
done
for t in 16 8 4 2; do
  timeout -k 10 300 python bench.py --workload kitti11 --host-threads $t --no-cpu --frames 127 --frame-cache gpurun_out/r03/cache > gpurun_out/r03/ht_kitti11_$t.json 2> gpurun_out/r03/ht_kitti11_$t.err || { tail -20 gpurun_out/r03/ht_kitti11_$t.err; exit 1; }
  python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/r03/ht_kitti11_$t.json') if l.startswith('{')][-1])
print('kitti11 host_threads=$t', round(j['value'],1), 'pnp_host ms', j['config']['stage_ms_timed_region'].get('pnp_host'))
PY
done
# the same with the minimal solves on the GPU (pnp_device = 1): what a rank with few cores gains
for t in 8 4 2; do
  timeout -k 10 300 python bench.py --host-threads $t --pnp-device 1 --no-cpu --no-upload-leg --no-latency --frame-cache gpurun_out/r03/cache > gpurun_out/r03/ht_kitti_dev_$t.json 2> gpurun_out/r03/ht_kitti_dev_$t.err || { tail -20 gpurun_out/r03/ht_kitti_dev_$t.err; exit 1; }
  python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/r03/ht_kitti_dev_$t.json') if l.startswith('{')][-1])
print('kitti, GPU minimal solves, host_threads=$t', round(j['value'],1), 'pnp_host ms', j['config']['stage_ms_timed_region'].get('pnp_host'))
PY
done
rm -rf gpurun_out/r03/cache
