"""GPU tests of the rows around the hot path: consecutive-frame (sequence) input, the KITTI directory runner and
the depth point-cloud export.  All go through the C ABI; expected values come from the pair-mode path (already pinned
against the oracle in test_gpu_parity.py) and from the oracle itself."""
import numpy as np
import pytest

from visualodometry_amd import abi, kitti, sequence, synth
from oracle import binding as ob

pytestmark = pytest.mark.gpu

W, H = 320, 96
PARAMS = dict(num_disparities=64, max_features=4096)


@pytest.fixture(scope="module")
def frames5():
    seq = synth.Sequence(seed=77, width=W, height=H)
    return seq, [seq.frame(i) for i in range(5)]


def test_sequence_mode_equals_pair_mode(frames5):
    """vo_set_sequence (every frame uploaded and pyramided once) gives bit-identical results to vo_set_images"""
    seq, frames = frames5
    n = len(frames) - 1
    ctx = abi.Context(abi.default_params(W, H, max_batch=n, **PARAMS))
    ctx.set_calib(seq.P0, seq.P1)
    lt1 = np.stack([frames[i][0] for i in range(n)]); rt1 = np.stack([frames[i][1] for i in range(n)])
    lt2 = np.stack([frames[i + 1][0] for i in range(n)]); rt2 = np.stack([frames[i + 1][1] for i in range(n)])
    ctx.set_images(lt1, rt1, lt2, rt2)
    Rt_a, info_a = ctx.step()
    pair = [(ctx.disparity16(b), ctx.features(b), ctx.matches(b), ctx.klt_legs(len(ctx.features(b)), b), ctx.object_points(b)) for b in range(n)]
    ctx.set_sequence(np.stack([f[0] for f in frames]), np.stack([f[1] for f in frames]))
    Rt_b, info_b = ctx.step()
    assert np.array_equal(Rt_a, Rt_b)
    for b in range(n):
        assert np.array_equal(ctx.disparity16(b), pair[b][0])
        assert np.array_equal(ctx.features(b), pair[b][1])
        for x, y in zip(ctx.matches(b), pair[b][2]):
            assert np.array_equal(x, y)
        for x, y in zip(ctx.klt_legs(len(ctx.features(b)), b), pair[b][3]):
            assert np.array_equal(x, y)
        for x, y in zip(ctx.object_points(b), pair[b][4]):
            assert np.array_equal(x, y)
        assert info_a[b].n_matched == info_b[b].n_matched and info_a[b].best_inliers == info_b[b].best_inliers
    # the whole batch's triangulated points with one copy: the same sets as frame by frame
    batch = ctx.object_points_batch()
    assert len(batch) == n
    for b in range(n):
        assert np.array_equal(batch[b], ctx.object_points(b)[0])
    capped = ctx.object_points_batch(cap=10)
    assert all(len(capped[b]) == min(10, len(batch[b])) and np.array_equal(capped[b], batch[b][:10]) for b in range(n))
    # the pyramid taps address (frame, side) slots in this mode: pair 1's LT1 pyramid is pair 0's LT2 pyramid
    a = ctx.tap(abi.TAP_PYR_LEVEL, 0, 2, 1)   # pair 0, image slot 2 (LT2), level 1
    b_ = ctx.tap(abi.TAP_PYR_LEVEL, 1, 0, 1)  # pair 1, image slot 0 (LT1), level 1
    assert np.array_equal(a, b_)
    # argument checks
    with pytest.raises(ValueError):
        ctx.set_sequence(np.stack([f[0] for f in frames])[:1], np.stack([f[1] for f in frames])[:1])
    with pytest.raises(abi.VoError):
        ctx.set_sequence(np.stack([f[0] for f in frames] * 2), np.stack([f[1] for f in frames] * 2))  # 9 pairs > max_batch
    ctx.close()


def test_kitti_directory_runner(frames5, tmp_path):
    """frames written as a KITTI sequence directory and run with kitti.run: the oracle's trajectory, the reference's RMSE"""
    seq, frames = frames5
    poses = [seq.pose(i)[:3, :] for i in range(len(frames))]
    kitti.write_sequence(str(tmp_path), frames, seq.P0, seq.P1, poses=poses)
    traj, err = kitti.run(str(tmp_path), poses_file=str(tmp_path / "poses.txt"), out_file=str(tmp_path / "out.txt"), batch=3, **PARAMS)
    op = ob.default_params(W, H, num_disparities=64)
    P0, P1 = kitti.read_calib(str(tmp_path / "calib.txt"))
    pose = np.eye(4)
    want = [np.zeros(3)]
    for i in range(len(frames) - 1):
        ob.step(frames[i][0], frames[i][1], frames[i + 1][0], frames[i + 1][1], op, P0, P1, pose)
        want.append(pose[:3, 3].copy())
    want = np.array(want)
    assert traj.shape == want.shape and np.abs(traj - want).max() < 1e-3
    gt = kitti.ground_truth(kitti.read_poses(str(tmp_path / "poses.txt")))
    assert abs(err - ob.trajectory_rmse(want, gt)) < 1e-3
    assert np.allclose(kitti.read_trajectory_xyz(str(tmp_path / "out.txt")), traj, rtol=1e-5, atol=1e-6)


def test_depth_point_cloud_export(frames5, tmp_path):
    seq, frames = frames5
    ctx = abi.Context(abi.default_params(W, H, **PARAMS))
    ctx.set_calib(seq.P0, seq.P1)
    ctx.set_images(frames[0][0], frames[0][1], frames[1][0], frames[1][1])
    ctx.compute_disparity()
    ctx.compute_depth_map()
    depth = ctx.depth(0)
    op = ob.default_params(W, H, num_disparities=64)
    d32 = ob.disp_to_float(ob.sgbm(frames[0][0], frames[0][1], op))
    fx = seq.P0[0, 0]
    want = ob.depth_map(d32, fx, seq.P1[0, 3] / (-1 * fx))
    assert np.array_equal(depth, want)
    written, header = kitti.write_depth_ply(str(tmp_path / "depth.ply"), depth, frames[0][0])
    lines = (tmp_path / "depth.ply").read_text().splitlines()
    assert lines[2] == "element vertex %d" % header and len(lines) == 10 + written
    # at this scale fx * baseline / 0.1 is below the 3000 threshold, so invalid pixels stay in the cloud (as in the reference)
    assert written == int(np.count_nonzero(depth.astype(np.float32) <= 3000)) and 0 < written <= W * H
    assert header == W * H - int(np.count_nonzero(depth.astype(np.int64) == int(depth.max())))
    r, c, z = lines[10].split()[:3]
    assert float(z) == pytest.approx(-depth[int(r), int(c)], rel=1e-5)
    # the same vertices compacted on the GPU: same records in the same order, and the same file body
    cloud = ctx.depth_cloud(0)
    z32 = depth.astype(np.float32)
    rows, cols = np.nonzero(~(z32 > 3000))
    assert len(cloud) == written
    assert np.array_equal(cloud["row"], rows) and np.array_equal(cloud["col"], cols)
    assert np.array_equal(cloud["neg_z"], -z32[rows, cols]) and np.array_equal(cloud["gray"], frames[0][0][rows, cols])
    kitti.write_cloud_ply(str(tmp_path / "cloud.ply"), cloud, n_vertices=header)
    assert (tmp_path / "cloud.ply").read_text() == (tmp_path / "depth.ply").read_text()
    # a threshold that rejects pixels (everything farther than 20 m): count-only query, then a capped copy
    near = abi.Context(abi.default_params(W, H, depth_thresh=20, **PARAMS))
    near.set_calib(seq.P0, seq.P1)
    near.set_images(frames[0][0], frames[0][1], frames[1][0], frames[1][1])
    near.compute_disparity()
    with pytest.raises(abi.VoError):
        near.depth_cloud(0)            # no depth map yet
    near.compute_depth_map()
    c20 = near.depth_cloud(0)
    keep = ~(near.depth(0).astype(np.float32) > 20)
    assert 0 < len(c20) == int(keep.sum()) < W * H
    assert np.array_equal(np.stack([c20["row"], c20["col"]], 1), np.argwhere(keep))
    near.close()
    ctx.close()


def test_chunk_of_a_sequence_equals_the_slice_of_the_whole(frames5):
    """chunked sharding: steps 1..2 computed alone are the same [R|t] as steps 1..2 of the whole sequence"""
    seq, frames = frames5
    ctx = abi.Context(abi.default_params(W, H, max_batch=4, **PARAMS))
    ctx.set_calib(seq.P0, seq.P1)
    whole, _ = sequence.run_sequence(ctx, frames)
    part, info = sequence.run_chunk(ctx, frames, 1, 3)
    assert part.shape == (2, 4, 4) and len(info) == 2
    assert np.array_equal(part, whole[1:3])
    ctx.close()


def test_stream_schedules_give_identical_results(monkeypatch):
    """A batch's tracking chain runs on a second stream behind the cost kernel by default; VO_SINGLE_STREAM keeps a context on
    one stream and VO_DUAL_STREAM starts the tracker at once (both read by vo_create).  Same kernels, same results."""
    W, H, nb = 640, 192, 5
    seq = synth.Sequence(seed=1011, width=W, height=H)
    frames = [seq.frame(i) for i in range(nb + 1)]
    left = np.stack([f[0] for f in frames]); right = np.stack([f[1] for f in frames])
    outs = []
    for env in (None, "VO_SINGLE_STREAM", "VO_DUAL_STREAM"):
        monkeypatch.delenv("VO_SINGLE_STREAM", raising=False); monkeypatch.delenv("VO_DUAL_STREAM", raising=False)
        if env:
            monkeypatch.setenv(env, "1")
        ctx = abi.Context(abi.default_params(W, H, max_batch=nb, num_disparities=64, max_features=4096))
        ctx.set_calib(seq.P0, seq.P1)
        ctx.set_sequence(left, right)
        for _ in range(2):   # a second step on the same context: the streams' events are reused
            Rt, info = ctx.step()
        outs.append((Rt.copy(), [ctx.disparity16(b).copy() for b in range(nb)], [ctx.matches(b)[0].copy() for b in range(nb)],
                     [(i.n_features, i.n_matched, i.n_object_points, i.best_inliers) for i in info]))
        ctx.close()
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and o[3] == outs[0][3]
        for b in range(nb):
            assert np.array_equal(o[1][b], outs[0][1][b]) and np.array_equal(o[2][b], outs[0][2][b])
