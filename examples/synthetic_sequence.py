#!/usr/bin/env python3
"""The reference's main() (Source/main.cpp:7-74) on a synthetic KITTI-shape sequence with exact ground truth:

    python examples/synthetic_sequence.py [--frames 200] [--batch 16] [--width 1241 --height 376] [--seed 1000]

Renders the frames (piece-wise planar textured scene, KITTI-00 intrinsics, 0.8 m forward per frame with a sinusoidal
yaw), runs every frame pair through the GPU path in batches, integrates poseRt = poseRt * Rt^-1 and prints the
translation RMSE against the ground truth the way the reference does (Source/VOClass.cpp:627-649)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualodometry_amd import abi, sequence, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--width", type=int, default=1241)
    ap.add_argument("--height", type=int, default=376)
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--out", default=None, help="pose dump (x y z per line), as the reference's Log/outputPoses.txt")
    a = ap.parse_args()
    seq = synth.Sequence(seed=a.seed, width=a.width, height=a.height)
    p = abi.default_params(a.width, a.height, max_batch=a.batch)
    ctx = abi.Context(p)
    ctx.set_calib(seq.P0, seq.P1)
    Rts, matched, inliers = [], [], []
    t_render = t_gpu = 0.0
    n_timed = 0
    last = None
    for s in range(0, a.frames - 1, a.batch):
        e = min(s + a.batch, a.frames - 1)
        t0 = time.perf_counter()
        frames = ([last] if last is not None else [seq.frame(s)]) + [seq.frame(i) for i in range(s + 1, e + 1)]
        last = frames[-1]
        t1 = time.perf_counter()
        Rt, info = sequence.run_sequence(ctx, frames, a.batch)
        t2 = time.perf_counter()
        t_render += t1 - t0
        if s > 0:   # the first batch also pays for loading the code object and the first allocations
            t_gpu += t2 - t1; n_timed += e - s
        Rts.extend(Rt); matched.extend(i.n_matched for i in info); inliers.extend(i.best_inliers for i in info)
        print("frames %d..%d done" % (s, e), file=sys.stderr, flush=True)
    ctx.close()
    traj, _ = sequence.integrate(Rts)
    # ground truth = camera positions relative to the first frame, as getGroundTruthPath keeps them
    gt = []
    T0 = np.vstack([seq.pose(0), [0, 0, 0, 1]]) if seq.pose(0).shape[0] == 3 else seq.pose(0)
    for i in range(a.frames):
        Ti = np.vstack([seq.pose(i), [0, 0, 0, 1]]) if seq.pose(i).shape[0] == 3 else seq.pose(i)
        gt.append((np.linalg.inv(T0) @ Ti)[:3, 3])
    gt = np.array(gt)
    err = sequence.trajectory_rmse(traj, gt)
    path = float(np.linalg.norm(np.diff(gt, axis=0), axis=1).sum())
    if a.out:
        from visualodometry_amd import kitti
        kitti.write_trajectory_xyz(a.out, traj)
    print("frames %d, path length %.1f m, matches per frame %.0f, inliers per frame %.0f" % (a.frames, path, np.mean(matched), np.mean(inliers)))
    print("Measured error: %g  (final position error %.3f m = %.3f %% of the path)" %
          (err, float(np.linalg.norm(traj[-1] - gt[-1])), 100 * float(np.linalg.norm(traj[-1] - gt[-1])) / max(path, 1e-9)))
    if n_timed:
        print("rendering %.1f s; GPU path incl. host-to-device uploads, after the first batch: %.2f s for %d frame pairs "
              "(%.0f frames/s with one context and no overlap)" % (t_render, t_gpu, n_timed, n_timed / t_gpu))


if __name__ == "__main__":
    main()
