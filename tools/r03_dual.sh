#!/bin/bash
# does running every batch's tracking chain on a second stream shorten the fill and drain of a short run?
set -o pipefail
mkdir -p gpurun_out/r03
for cfg in "base:" "dual:VO_DUAL_STREAM=1" "dualq8:VO_DUAL_STREAM=1 GPU_MAX_HW_QUEUES=8"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  for steps in 20 20 300; do
    env $envs timeout -k 10 280 python bench.py --steps $steps --warmup 5 --no-cpu --no-upload-leg --no-latency > gpurun_out/r03/dual_${tag}_$steps.json 2> gpurun_out/r03/dual_${tag}_$steps.err || { tail -5 gpurun_out/r03/dual_${tag}_$steps.err; exit 1; }
    python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r03/dual_${tag}_$steps.json") if l.startswith("{")][-1])
print("$tag", $steps, round(j["value"],1), round(j["ms_per_step"],3))
PY
  done
done
