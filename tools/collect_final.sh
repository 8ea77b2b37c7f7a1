#!/bin/bash
# bench lines of every workload at the final kernels (no profiler): into gpurun_out/final/
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/final; mkdir -p $out
C="--frame-cache /tmp/vo_frames"
timeout -k 10 300 python3 $root/bench.py $C > $out/bench_default.json 2> $out/bench_default.err
timeout -k 10 300 python3 $root/bench.py $C --steps 20 --warmup 5 > $out/bench_steps20.json 2> $out/bench_steps20.err
# (the 4K run goes last: the run that follows it is slow, whatever it is: 4 450 frames/s for the default workload
# against 5 950 before it, its host part twice as long, while the 132 GB the 4K contexts held are being released)
timeout -k 10 300 python3 $root/bench.py --workload kitti11 --no-cpu --frames 127 $C > $out/bench_kitti11.json 2> $out/bench_kitti11.err
timeout -k 10 300 python3 $root/bench.py --workload dense --steps 40 $C > $out/bench_dense.json 2> $out/bench_dense.err
timeout -k 10 400 python3 $root/bench.py --workload 4k --steps 50 $C > $out/bench_4k.json 2> $out/bench_4k.err
python3 - <<P
import json
for f in ("default","steps20","dense","4k","kitti11"):
    d=json.loads(open("$out/bench_%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(d["value"],1), round(d["ms_per_step"],3), d["config"].get("latency_ms_one_pair_alone"), d["config"].get("value_with_upload"))
P
