#!/bin/bash
# A/B of kernel variants on a GPU box: tools/ab.sh <rounds> <lib.so | base> ...   (libraries under gpurun_variants/)
# Prints, per run, the bench value and the one-batch-in-flight stage times; extra bench flags come from $AB_FLAGS.
root=$(cd "$(dirname "$0")/.." && pwd)
rounds=$1; shift
for i in $(seq $rounds); do
  for v in "$@"; do
    if [ "$v" = base ]; then unset VO_AMD_LIB; else export VO_AMD_LIB=$root/gpurun_variants/$v; fi
    timeout -k 10 200 python3 $root/bench.py --frame-cache /tmp/vo_frames --no-cpu --no-upload-leg --steps ${AB_STEPS:-100} $AB_FLAGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['config']['stage_ms_one_batch_in_flight']
print('$v', round(d['value'],1), ' '.join('%s=%.3f'%(k.replace('sgbm_',''),s[k]) for k in ('sgbm_cost_vpath','sgbm_hpath_ckpt','sgbm_hpath_resolve','lk','pyramid','fast','sgbm_prefilter','sgbm_lrcheck_median','klt_filter')))" || exit 1
  done
done
