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


def test_frame4k_256_disparities_against_the_oracle_fixture():
    """the 4K configuration at full size (3840x2160, 256 disparity levels: stripes of 540 rows with 60 warm-up rows, 16
    disparities per lane): disparity before and after the median and the FAST list bit-exact against the oracle's hashes
    (tests/golden/frame4k_d256.npz, made by tests/golden/make_frame4k.py), and the whole main-loop step on that frame pair"""
    fx = np.load(os.path.join(GOLDEN, "frame4k_d256.npz"))
    W, H, D, seed = (int(fx[k]) for k in ("width", "height", "num_disparities", "seed"))
    scene = synth.pixel_matched_scene(W)
    frames = render_frames(seed, W, H, 2, **scene)
    got_hash = np.array([[h64(f[0]), h64(f[1])] for f in frames], dtype=np.uint64)
    assert np.array_equal(got_hash, fx["frame_hash"]), "the rendered input frames differ from those the fixture was made from"
    seq = synth.Sequence(seed=seed, width=W, height=H, **scene)
    ctx = abi.Context(abi.default_params(W, H, max_batch=1, num_disparities=D, max_features=98304))
    ctx.set_calib(seq.P0, seq.P1)
    ctx.set_images(frames[0][0], frames[0][1], frames[1][0], frames[1][1])
    Rt, info = ctx.step()
    d16 = ctx.disparity16(0)
    pre = ctx.tap(abi.TAP_DISP_PREMEDIAN, 0)
    s = fx["samples"]
    wrong = [(int(y), int(x), int(d16[y, x]), int(w)) for y, x, w, _ in s if d16[y, x] != w]
    assert not wrong, "sampled disparities differ from the oracle's (y, x, got, want): %s" % wrong[:8]
    wrong = [(int(y), int(x), int(pre[y, x]), int(w)) for y, x, _, w in s if pre[y, x] != w]
    assert not wrong, "sampled pre-median disparities differ from the oracle's: %s" % wrong[:8]
    assert h64(pre) == int(fx["premedian_hash"])
    assert h64(d16) == int(fx["disp_hash"])
    assert int((d16[:, D:] < 0).sum()) == int(fx["n_invalid"])
    assert h64(ctx.features(0)) == int(fx["fast_hash"])
    f = info[0]
    assert (f.n_features, f.n_matched, f.n_object_points) == (int(fx["n_fast"]), int(fx["n_matched"]), int(fx["n_obj"]))
    assert (f.ok, f.iters_run, f.best_inliers) == (int(fx["pnp_ok"]), int(fx["pnp_iters"]), int(fx["pnp_inliers"]))
    t1, t2 = ctx.matches(0)
    if h64(np.concatenate([t1, t2])) != int(fx["match_hash"]):   # not bit-identical: the 0.01 px tolerance on the kept sample
        k = fx["match_sample"].shape[1]
        assert np.abs(t1[:k] - fx["match_sample"][0]).max() <= 0.01 and np.abs(t2[:k] - fx["match_sample"][1]).max() <= 0.01
    want = fx["Rt"]
    assert rotation_angle(want[:3, :3], Rt[0][:3, :3]) <= 1e-4 and np.abs(Rt[0][:3, 3] - want[:3, 3]).max() <= 1e-3
    assert f.n_matched >= 0.7 * f.n_features       # the scene keeps the matching circle as closed as at KITTI shape
    ctx.close()
