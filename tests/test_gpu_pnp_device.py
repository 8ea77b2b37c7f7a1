"""RANSAC minimal solves on the GPU (vo_params.pnp_device = 1) against the host solves of the same library and against
the oracle: same subsets, same sequential accept rule, the 5-point EPnP arithmetic of epnp5.h in f64 on both sides.  A
hypothesis may differ from the host's (device hypot / acos / sin / cos differ in the last bits, and a five-point solve can
amplify that), so the comparisons state tolerances; on the small batches of the first test the two modes agree bit for bit
in practice, over the 200-frame sequence 196 of 199 steps do."""
import numpy as np
import pytest

from visualodometry_amd import abi, synth
from oracle import binding as ob

pytestmark = pytest.mark.gpu

W, H = 640, 192


def _run(frames, seq, nb, dev, **kw):
    p = abi.default_params(W, H, max_batch=nb, num_disparities=64, max_features=4096, pnp_device=dev, **kw)
    ctx = abi.Context(p)
    ctx.set_calib(seq.P0, seq.P1)
    ctx.set_sequence(np.stack([f[0] for f in frames]), np.stack([f[1] for f in frames]))
    Rt, info = ctx.step()
    out = (Rt.copy(), [(i.ok, i.iters_run, i.best_inliers, i.n_object_points, i.final_from_refine) for i in info])
    ctx.close()
    return out


def _rot_angle(Ra, Rb):
    c = (np.trace(Ra.T @ Rb) - 1) * 0.5
    return float(np.arccos(min(1.0, max(-1.0, c))))


@pytest.mark.parametrize("kw", [dict(), dict(ransac_iters=512, ransac_chunk=512)], ids=["rounds_of_8", "one_round_of_512_mask_pass"])
def test_gpu_minimal_solves_match_host_and_oracle(kw):
    nb = 3
    seq = synth.Sequence(seed=1007, width=W, height=H)
    frames = [seq.frame(i) for i in range(nb + 1)]
    Rt0, i0 = _run(frames, seq, nb, 0, **kw)
    Rt1, i1 = _run(frames, seq, nb, 1, **kw)
    okw = {k: v for k, v in kw.items() if k != "ransac_chunk"}
    op = ob.default_params(W, H, num_disparities=64, **okw)
    for b in range(nb):
        assert i0[b][0] == i1[b][0] == 1 and i0[b][3] == i1[b][3] and i0[b][4] == i1[b][4] == 1
        assert abs(i0[b][2] - i1[b][2]) <= max(2, 0.005 * i0[b][3]), (b, i0[b], i1[b])
        assert _rot_angle(Rt0[b][:3, :3], Rt1[b][:3, :3]) <= 1e-4 and np.abs(Rt0[b][:3, 3] - Rt1[b][:3, 3]).max() <= 1e-3
        want, oinfo = ob.step(frames[b][0], frames[b][1], frames[b + 1][0], frames[b + 1][1], op, seq.P0, seq.P1, np.eye(4))
        assert _rot_angle(want[:3, :3], Rt1[b][:3, :3]) <= 1e-4 and np.abs(want[:3, 3] - Rt1[b][:3, 3]).max() <= 1e-3
        assert abs(oinfo.pnp.best_inliers - i1[b][2]) <= max(2, 0.005 * i1[b][3])


def test_pnp_device_values_outside_0_1_are_refused():
    with pytest.raises(abi.VoError) as e:
        abi.Context(abi.default_params(W, H, pnp_device=2))
    assert e.value.code == -2


def test_gpu_minimal_solves_over_the_200_frame_sequence():
    """All 199 steps of the KITTI-shape sequence with the minimal solves on the GPU, against the oracle-made fixture
    (tests/golden/kitti200.npz): relative pose within 1e-4 rad / 1e-3 m, integrated trajectory within 0.1 % of the path length.
    The RANSAC outcome itself is the host's on nearly every step but not on all: a five-point solve can amplify a last-bit
    difference of the device's hypot / acos (its null space is two-dimensional) into a visibly different hypothesis, whose
    inlier set then differs by a few borderline points (step 119: 4 025 inliers against 4 028 of 4 053).  Stated here:
    inlier counts within 0.5 % of the correspondences, and at least 95 % of the steps with the oracle's exact counts."""
    import os
    from fullsize_util import render_frames, rotation_angle
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti200.npz"))
    Wk, Hk, D, NF, seed = (int(fx[k]) for k in ("width", "height", "num_disparities", "n_frames", "seed"))
    frames = render_frames(seed, Wk, Hk, NF)
    left = np.stack([f[0] for f in frames]); right = np.stack([f[1] for f in frames])
    seq = synth.Sequence(seed=seed, width=Wk, height=Hk)
    B = 42
    ctx = abi.Context(abi.default_params(Wk, Hk, max_batch=B, num_disparities=D, max_features=16384, pnp_device=1))
    ctx.set_calib(seq.P0, seq.P1)
    pose = np.eye(4); traj = np.zeros((NF, 3)); bad = []; same = 0
    for s0 in range(0, NF - 1, B):
        n = min(B, NF - 1 - s0)
        ctx.set_sequence(left[s0:s0 + n + 1], right[s0:s0 + n + 1])
        Rt, info = ctx.step()
        for b in range(n):
            i = s0 + b
            f = info[b]
            if f.ok != int(fx["pnp_ok"][i]) or abs(f.best_inliers - int(fx["pnp_inliers"][i])) > 0.005 * f.n_object_points:
                bad.append((i, "RANSAC %s" % ((f.ok, f.iters_run, f.best_inliers),)))
            same += (f.iters_run, f.best_inliers) == (int(fx["pnp_iters"][i]), int(fx["pnp_inliers"][i]))
            want = fx["Rt"][i]
            rot, dt = rotation_angle(want[:3, :3], Rt[b][:3, :3]), float(np.abs(Rt[b][:3, 3] - want[:3, 3]).max())
            if rot > 1e-4 or dt > 1e-3:
                bad.append((i, "pose off by %.2e rad, %.2e m" % (rot, dt)))
            abi.integrate_pose(pose, Rt[b])
            traj[i + 1] = pose[:3, 3]
    ctx.close()
    assert not bad, bad[:10]
    assert float(np.abs(traj - fx["trajectory"]).max()) <= 1e-3 * float(fx["path_length"])
    assert same >= 0.95 * (NF - 1), same
    print("GPU minimal solves: %d of %d steps with the oracle's exact iteration and inlier counts" % (same, NF - 1))


def test_gpu_minimal_solves_in_a_mixed_batch():
    """A batch (shorter than the context's capacity) whose frames reach every branch around the rounds: many correspondences
    (RANSAC rounds, solved on the GPU), exactly five and exactly four (one direct solve on the host, no rounds) and three
    (refused).  The frames without rounds must come out exactly as with host solves, the one with rounds within tolerance."""
    from scipy.spatial.transform import Rotation
    P0, P1 = synth.calib(W, H)
    fx, cx, cy = P0[0, 0], P0[0, 2], P0[1, 2]
    rvec, tvec = np.array([0.01, -0.02, 0.005]), np.array([0.05, -0.01, -0.8])
    vv, uu = np.mgrid[0:H, 0:W]
    depth = 6.0 + 0.03 * uu + 0.05 * vv + 2.0 * np.sin(uu / 37.0) * np.cos(vv / 23.0)
    rng = np.random.RandomState(11)
    sets = [np.stack([rng.uniform(20, W - 20, 60), rng.uniform(10, H - 10, 60)], axis=1),
            np.array([[100.3, 40.7], [500.2, 60.1], [320.9, 150.4], [200.5, 100.5]]),
            np.array([[100.3, 40.7], [500.2, 60.1], [320.9, 150.4], [200.5, 100.5], [420.25, 33.5]]),
            np.array([[100.3, 40.7], [500.2, 60.1], [320.9, 150.4]])]
    results = []
    for dev in (0, 1):
        ctx = abi.Context(abi.default_params(W, H, max_batch=6, num_disparities=64, max_features=4096, pnp_device=dev))
        ctx.set_calib(P0, P1)
        blank = np.zeros((len(sets), H, W), dtype=np.uint8)
        ctx.set_images(blank, blank, blank, blank)
        ctx.compute_disparity(); ctx.compute_depth_map()    # a batch's maps exist before a caller replaces them frame by frame
        for b, pts in enumerate(sets):
            t1 = pts.astype(np.float32)
            ctx.set_depth(depth, b)
            obj, _, _ = ob.triangulate(t1, t1, depth, P0)
            Xc = Rotation.from_rotvec(rvec).apply(obj.astype(np.float64)) + tvec
            t2 = np.stack([fx * Xc[:, 0] / Xc[:, 2] + cx, fx * Xc[:, 1] / Xc[:, 2] + cy], axis=1)
            t2 += np.random.RandomState(b).normal(0, 0.2, t2.shape) if b == 0 else 0.0
            ctx.set_matches(t1, t2.astype(np.float32), b)
        ctx.triangulate()
        Rt = np.zeros((len(sets), 4, 4)); info = (abi.FrameInfo * len(sets))()
        rc = ctx._L.vo_estimate_motion(ctx._h, Rt.ctypes.data_as(abi.C.c_void_p), info)
        assert rc == -7     # VO_ERR_NOT_ENOUGH_POINTS: the three-point frame; every frame's outputs are filled all the same
        results.append((Rt.copy(), [(i.ok, i.iters_run, i.best_inliers, i.n_object_points) for i in info]))
        ctx.close()
    (Rt0, i0), (Rt1, i1) = results
    assert [x[3] for x in i0] == [60, 4, 5, 3] and [x[0] for x in i0] == [1, 1, 1, 0]
    assert i0[1:] == i1[1:] and np.array_equal(Rt0[1:], Rt1[1:])            # no rounds: the host solved them either way
    assert i1[0][0] == 1 and abs(i0[0][2] - i1[0][2]) <= 2
    assert _rot_angle(Rt0[0][:3, :3], Rt1[0][:3, :3]) <= 1e-4 and np.abs(Rt0[0][:3, 3] - Rt1[0][:3, 3]).max() <= 1e-3
    assert _rot_angle(Rt1[0][:3, :3], Rotation.from_rotvec(rvec).as_matrix()) <= 2e-3 and np.abs(Rt1[0][:3, 3] - tvec).max() <= 2e-2
