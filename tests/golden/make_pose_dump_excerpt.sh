#!/bin/sh
# Cuts the excerpt of the reference's own pose dump (its run on KITTI sequence 00; a data file, "x y z" per line as
# written by Source/VOClass.cpp:607-621) that tests/test_kitti_io.py replays through the trajectory reader and writer:
# the first 200 and the last 20 of its 4541 lines.
set -e
src=/root/reference/Log/outputPoses.txt
out=$(dirname "$0")/reference_pose_dump_excerpt.txt
{ head -n 200 "$src"; tail -n 20 "$src"; } > "$out"
wc -l "$out"
