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
