#!/bin/bash
# A/B of a batch's stream schedule on one GPU: everything on one stream with 4 hardware queues (the schedule up to now),
# the tracking chain on a second stream started at once, and the default (second stream, tracker behind the cost kernel,
# i.e. beside the horizontal path passes) with 4 and with 8 hardware queues (bench.py sets GPU_MAX_HW_QUEUES itself: --hw-queues).
# Prints frames/s per configuration.
set -e
out=gpurun_out/r03_schedule; mkdir -p $out
run() { # name, env assignments, bench arguments
  env $2 timeout -k 10 200 python bench.py $3 --no-latency --no-collective --no-cpu > $out/$1.json 2> $out/$1.err
  python - <<P
import json
d=json.loads(open("$out/$1.json").read().strip().splitlines()[-1])
print("$1: %.1f frames/s, %.3f ms/step" % (d["value"], d["ms_per_step"]))
P
}
for rep in 1 2; do
  run kitti_single_q4_r$rep "VO_SINGLE_STREAM=1" "--steps 300 --warmup 10 --hw-queues 4"
  run kitti_atonce_q8_r$rep "VO_DUAL_STREAM=1" "--steps 300 --warmup 10 --hw-queues 8"
  run kitti_default_q4_r$rep "" "--steps 300 --warmup 10 --hw-queues 4"
  run kitti_default_q8_r$rep "" "--steps 300 --warmup 10 --hw-queues 8"
done
run steps20_single_q4 "VO_SINGLE_STREAM=1" "--steps 20 --warmup 3 --hw-queues 4"
run steps20_default_q8 "" "--steps 20 --warmup 3 --hw-queues 8"
run 4k_single_q4 "VO_SINGLE_STREAM=1" "--workload 4k --steps 30 --warmup 3 --hw-queues 4"
run 4k_default_q8 "" "--workload 4k --steps 30 --warmup 3 --hw-queues 8"
run dense_single_q4 "VO_SINGLE_STREAM=1" "--workload dense --steps 30 --warmup 3 --hw-queues 4"
run dense_default_q8 "" "--workload dense --steps 30 --warmup 3 --hw-queues 8"
run kitti11_single_q8 "VO_SINGLE_STREAM=1" "--workload kitti11 --hw-queues 8"
run kitti11_default_q8 "" "--workload kitti11 --hw-queues 8"
