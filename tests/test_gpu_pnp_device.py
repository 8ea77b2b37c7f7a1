"""RANSAC minimal solves on the GPU (vo_params.pnp_device = 1) against the host solves of the same library and against
the oracle: same subsets, same sequential accept rule, the 5-point EPnP arithmetic of epnp5.h in f64 on both sides.  A
hypothesis may differ from the host's in its last bits (device hypot / acos / sin / cos), so the comparison states
tolerances: inlier counts within 2 of each other, relative pose within 1e-6 rad / 1e-6 m of the host mode and within the
task's 1e-4 rad / 1e-3 m of the oracle.  In practice the two modes agree bit for bit."""
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
        assert abs(i0[b][2] - i1[b][2]) <= 2, (b, i0[b], i1[b])
        assert _rot_angle(Rt0[b][:3, :3], Rt1[b][:3, :3]) <= 1e-6 and np.abs(Rt0[b][:3, 3] - Rt1[b][:3, 3]).max() <= 1e-6
        want, oinfo = ob.step(frames[b][0], frames[b][1], frames[b + 1][0], frames[b + 1][1], op, seq.P0, seq.P1, np.eye(4))
        assert _rot_angle(want[:3, :3], Rt1[b][:3, :3]) <= 1e-4 and np.abs(want[:3, 3] - Rt1[b][:3, 3]).max() <= 1e-3
        assert abs(oinfo.pnp.best_inliers - i1[b][2]) <= 2


def test_pnp_device_values_outside_0_1_are_refused():
    with pytest.raises(abi.VoError) as e:
        abi.Context(abi.default_params(W, H, pnp_device=2))
    assert e.value.code == -2
