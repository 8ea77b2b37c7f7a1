"""Full-size parity on the GPU against oracle-made fixtures: the oracle cannot travel at these sizes in test time, its
outputs (as hashes, counts and poses) can.

* the WHOLE 200-frame KITTI-shape sequence of the single-GPU configuration (tests/golden/kitti200.npz, made by
  tests/golden/make_kitti200.py): on every one of the 199 main-loop steps (Source/main.cpp:36-60) the s16 disparity map and
  the FAST list are bit-exact, the counts the reference logs are equal, the circular-matching survivors are within 0.01 px,
  the relative pose within 1e-4 rad / 1e-3 m, and the integrated trajectory (Source/VOClass.cpp:587,597-599) within 0.1 % of
  the path length of the oracle's, with the same RMSE against the generating poses (Source/VOClass.cpp:627-649).

The fixtures are ORACLE-MADE (the reference holds none for this path): they pin the GPU path to the oracle at full size.
"""
import os

import numpy as np
import pytest

from visualodometry_amd import abi, synth
from fullsize_util import h64, render_frames, rotation_angle

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_kitti200_every_frame_against_the_oracle_fixture():
    fx = np.load(os.path.join(GOLDEN, "kitti200.npz"))
    W, H, D, NF, seed = (int(fx[k]) for k in ("width", "height", "num_disparities", "n_frames", "seed"))
    frames = render_frames(seed, W, H, NF)
    got_hash = np.array([[h64(f[0]), h64(f[1])] for f in frames], dtype=np.uint64)
    assert np.array_equal(got_hash, fx["frame_hash"]), "the rendered input frames differ from those the fixture was made from"
    left = np.stack([f[0] for f in frames])
    right = np.stack([f[1] for f in frames])
    seq = synth.Sequence(seed=seed, width=W, height=H)
    B = 42
    ctx = abi.Context(abi.default_params(W, H, max_batch=B, num_disparities=D, max_features=16384))
    ctx.set_calib(seq.P0, seq.P1)
    ns = NF - 1
    bad = []
    worst_rot = worst_t = worst_px = 0.0
    pose = np.eye(4)
    traj = np.zeros((NF, 3))
    nsample = fx["match_sample"].shape[2]
    for s0 in range(0, ns, B):
        n = min(B, ns - s0)
        ctx.set_sequence(left[s0:s0 + n + 1], right[s0:s0 + n + 1])   # every frame uploaded and pyramided once
        Rt, info = ctx.step()
        for b in range(n):
            i = s0 + b
            if h64(ctx.disparity16(b)) != int(fx["disp_hash"][i]):
                bad.append((i, "disparity"))
            if h64(ctx.features(b)) != int(fx["fast_hash"][i]):
                bad.append((i, "FAST list"))
            f = info[b]
            if (f.n_features, f.n_matched, f.n_object_points) != (int(fx["n_fast"][i]), int(fx["n_matched"][i]), int(fx["n_obj"][i])):
                bad.append((i, "counts %s" % ((f.n_features, f.n_matched, f.n_object_points),)))
            if (f.ok, f.iters_run, f.best_inliers) != (int(fx["pnp_ok"][i]), int(fx["pnp_iters"][i]), int(fx["pnp_inliers"][i])):
                bad.append((i, "RANSAC %s" % ((f.ok, f.iters_run, f.best_inliers),)))
            t1, t2 = ctx.matches(b)
            if h64(np.concatenate([t1, t2])) != int(fx["match_hash"][i]):   # not bit-identical: the 0.01 px tolerance on the kept sample
                k = min(nsample, len(t1))
                e = max(np.abs(t1[:k] - fx["match_sample"][i, 0, :k]).max(), np.abs(t2[:k] - fx["match_sample"][i, 1, :k]).max()) if k else 0.0
                worst_px = max(worst_px, float(e))
                if e > 0.01:
                    bad.append((i, "tracks off by %.4f px" % e))
            want = fx["Rt"][i]
            rot, dt = rotation_angle(want[:3, :3], Rt[b][:3, :3]), float(np.abs(Rt[b][:3, 3] - want[:3, 3]).max())
            worst_rot, worst_t = max(worst_rot, rot), max(worst_t, dt)
            if rot > 1e-4 or dt > 1e-3:
                bad.append((i, "pose off by %.2e rad, %.2e m" % (rot, dt)))
            abi.integrate_pose(pose, Rt[b])
            traj[i + 1] = pose[:3, 3]
    ctx.close()
    assert not bad, "%d of %d steps differ from the oracle fixture: %s" % (len({i for i, _ in bad}), ns, bad[:12])
    path = float(fx["path_length"])
    ate = float(np.abs(traj - fx["trajectory"]).max())
    assert ate <= 1e-3 * path, "trajectory off the oracle's by %.4f m over a %.1f m path" % (ate, path)
    gt = np.stack([seq.pose(i)[:, 3] for i in range(NF)])
    from visualodometry_amd import sequence
    rmse = sequence.trajectory_rmse(traj, gt)
    assert abs(rmse - float(fx["rmse_vs_truth"])) <= 1e-3 * path
    print("kitti200: %d steps bit-exact (disparity, FAST); worst pose difference %.1e rad / %.1e m, worst track %.1e px, "
          "trajectory within %.1e m of the oracle's, RMSE against the generating poses %.3f m" % (ns, worst_rot, worst_t, worst_px, ate, rmse))
