"""Regenerates tests/golden/kitti200.npz: the CPU oracle's outputs on the WHOLE 200-frame KITTI-shape synthetic sequence
(1241x376, 96 disparities, seed 1000: the single-GPU configuration of the benchmark), reduced to what fits a fixture.

Per main-loop step (199 of them, Source/main.cpp:36-60): a 64-bit hash of the s16 disparity map, of the FAST keypoint list
and of the circular-matching survivors, the counts the reference logs (features, matches, points with depth), the RANSAC
iteration and inlier counts, the relative pose [R|t] (f64), the first matched point pairs, and the integrated trajectory
(Source/VOClass.cpp:587,597-599).  Hashes of the rendered input frames ride along so that a test can tell a rendering
difference from a kernel difference.

The vectors are ORACLE-MADE (this repository's restatement of OpenCV 4.5.4; the reference holds no fixture for this path), so
they pin the GPU path to the oracle over every frame of the sequence, not the oracle to OpenCV.

Run from the repository root (about 4 minutes on 8 cores):  python tests/golden/make_kitti200.py
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from visualodometry_amd import synth  # noqa: E402

SEED, W, H, D, NFRAMES, NSAMPLE = 1000, 1241, 376, 96, 200, 32


def h64(a):
    """64-bit BLAKE2b digest of an array's bytes (C order) as an unsigned integer."""
    return int.from_bytes(hashlib.blake2b(np.ascontiguousarray(a).tobytes(), digest_size=8).digest(), "little")


def _render(i):
    return synth.Sequence(seed=SEED, width=W, height=H).frame(i)


def render_all(n, workers):
    if workers <= 1:
        return [_render(i) for i in range(n)]
    import multiprocessing as mp
    with mp.get_context("fork").Pool(workers) as pool:
        return pool.map(_render, range(n), chunksize=1)


def main():
    from oracle import binding as ob
    t0 = time.time()
    frames = render_all(NFRAMES, min(8, os.cpu_count() or 1))
    print("rendered %d frames in %.0f s" % (NFRAMES, time.time() - t0), flush=True)
    seq = synth.Sequence(seed=SEED, width=W, height=H)
    ob.set_threads(min(8, os.cpu_count() or 1))
    p = ob.default_params(W, H, num_disparities=D)
    ns = NFRAMES - 1
    out = dict(
        frame_hash=np.array([[h64(f[0]), h64(f[1])] for f in frames], dtype=np.uint64),
        disp_hash=np.zeros(ns, dtype=np.uint64), fast_hash=np.zeros(ns, dtype=np.uint64), match_hash=np.zeros(ns, dtype=np.uint64),
        n_fast=np.zeros(ns, dtype=np.int32), n_matched=np.zeros(ns, dtype=np.int32), n_obj=np.zeros(ns, dtype=np.int32),
        pnp_ok=np.zeros(ns, dtype=np.int32), pnp_iters=np.zeros(ns, dtype=np.int32), pnp_inliers=np.zeros(ns, dtype=np.int32),
        n_invalid=np.zeros(ns, dtype=np.int32),
        Rt=np.zeros((ns, 4, 4)), match_sample=np.zeros((ns, 2, NSAMPLE, 2), dtype=np.float32),
        trajectory=np.zeros((NFRAMES, 3)))
    pose = np.eye(4)
    t0 = time.time()
    for i in range(ns):
        a, b = frames[i], frames[i + 1]
        Rt, info, d16, kp, t1, t2 = ob.step_ex(a[0], a[1], b[0], b[1], p, seq.P0, seq.P1, pose)
        out["disp_hash"][i] = h64(d16)
        out["fast_hash"][i] = h64(kp)
        out["match_hash"][i] = h64(np.concatenate([t1, t2]))
        out["n_fast"][i], out["n_matched"][i], out["n_obj"][i] = info.n_fast, info.n_matched, info.n_obj
        out["pnp_ok"][i], out["pnp_iters"][i], out["pnp_inliers"][i] = info.pnp.ok, info.pnp.iters_run, info.pnp.best_inliers
        out["n_invalid"][i] = int((d16 < 0).sum())
        out["Rt"][i] = Rt
        k = min(NSAMPLE, len(t1))
        out["match_sample"][i, 0, :k] = t1[:k]
        out["match_sample"][i, 1, :k] = t2[:k]
        out["trajectory"][i + 1] = pose[:3, 3]
        if i % 20 == 0:
            print("step %d: %d features, %d matches, %d inliers, %.0f s" % (i, info.n_fast, info.n_matched, info.pnp.best_inliers, time.time() - t0), flush=True)
    gt = np.stack([seq.pose(i)[:, 3] for i in range(NFRAMES)])
    rmse = ob.trajectory_rmse(out["trajectory"], gt)
    path = float(np.linalg.norm(np.diff(gt, axis=0), axis=1).sum())
    print("oracle trajectory: RMSE %.4f m against the generating poses over a %.1f m path" % (rmse, path))
    np.savez_compressed(os.path.join(HERE, "kitti200.npz"), seed=SEED, width=W, height=H, num_disparities=D, n_frames=NFRAMES,
                        rmse_vs_truth=rmse, path_length=path, **out)


if __name__ == "__main__":
    main()
