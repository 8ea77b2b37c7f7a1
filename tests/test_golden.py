"""Committed golden vectors (tests/golden/*.npz, produced by tests/golden/make_golden.py with the CPU oracle).

CPU part: the oracle still reproduces them (the restatement has not drifted).
GPU part: the HIP path reproduces them through the C ABI, without the oracle in the loop.
"""
import os

import numpy as np
import pytest

from visualodometry_amd import abi
from oracle import binding as ob

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STEREO = ("stereo_320x96_d64", "stereo_250x75_d32")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.mark.parametrize("name", STEREO)
def test_oracle_reproduces_golden(name):
    g = load(name)
    H, W = g["lt1"].shape
    p = ob.default_params(W, H, num_disparities=int(g["num_disparities"]))
    d16, pre = ob.sgbm(g["lt1"], g["rt1"], p, with_premedian=True)
    assert np.array_equal(d16, g["disp16"]) and np.array_equal(pre, g["disp_premedian"])
    kp = ob.fast(g["lt1"])
    assert np.array_equal(kp, g["keypoints"])
    t1, t2, legs, st = ob.match_klt(g["lt1"], g["rt1"], g["lt2"], g["rt2"], kp, p, with_legs=True)
    assert np.array_equal(st, g["leg_status"]) and np.array_equal(t1, g["match_t1"]) and np.array_equal(t2, g["match_t2"])
    ok, rvec, tvec, mask, info = ob.solve_pnp_ransac(g["object_points"], g["object_img2"], g["P0"], p)
    assert ok == 1 and np.array_equal(mask, g["inlier_mask"]) and info.iters_run == int(g["pnp_iters"])
    assert np.allclose(rvec, g["rvec"], atol=1e-12) and np.allclose(tvec, g["tvec"], atol=1e-12)
    assert np.array_equal(ob.pyrdown(g["lt1"]), g["pyr1"]) and np.array_equal(ob.scharr(g["pyr1"]), g["scharr1"])


def test_oracle_reproduces_hard_sgbm_and_rng():
    g = load("sgbm_hard_200x60_d32")
    H, W = g["left"].shape
    p = ob.default_params(W, H, num_disparities=32)
    d16, pre = ob.sgbm(g["left"], g["right"], p, with_premedian=True)
    assert np.array_equal(d16, g["disp16"]) and np.array_equal(pre, g["disp_premedian"])
    assert (g["disp16"][:, 32:] < 0).sum() > 100          # left-right failures are present
    assert (g["disp16"] != g["disp16_scalar_ties"]).sum() > 1000  # and exact ties, resolved differently by the two rules
    r = load("rng_subsets")
    for c in ("6", "100", "4000"):
        assert np.array_equal(ob.rng_subsets(int(c), 100), r[c])


def test_oracle_reproduces_kitti200_steps():
    """full-length fixture (tests/golden/make_kitti200.py): the oracle still gives its hashes, counts and poses on the steps
    tried here (the first, and one from the middle), and the renderer still gives the frames it was made from"""
    from visualodometry_amd import synth
    from fullsize_util import h64
    g = load("kitti200")
    W, H, D, seed = (int(g[k]) for k in ("width", "height", "num_disparities", "seed"))
    seq = synth.Sequence(seed=seed, width=W, height=H)
    p = ob.default_params(W, H, num_disparities=D)
    ob.set_threads(min(8, os.cpu_count() or 1))
    for i in (0, 117):
        a, b = seq.frame(i), seq.frame(i + 1)
        assert [h64(a[0]), h64(a[1])] == [int(v) for v in g["frame_hash"][i]]
        pose = np.eye(4)
        Rt, info, d16, kp, t1, t2 = ob.step_ex(a[0], a[1], b[0], b[1], p, seq.P0, seq.P1, pose)
        assert h64(d16) == int(g["disp_hash"][i]) and h64(kp) == int(g["fast_hash"][i])
        assert h64(np.concatenate([t1, t2])) == int(g["match_hash"][i])
        assert (info.n_fast, info.n_matched, info.n_obj) == (int(g["n_fast"][i]), int(g["n_matched"][i]), int(g["n_obj"][i]))
        assert (info.pnp.iters_run, info.pnp.best_inliers) == (int(g["pnp_iters"][i]), int(g["pnp_inliers"][i]))
        assert np.allclose(Rt, g["Rt"][i], atol=1e-12)
    ob.set_threads(1)
    # the integrated trajectory of the fixture is the ordered product of its own relative poses (Source/VOClass.cpp:587)
    pose = np.eye(4)
    for i in range(len(g["Rt"])):
        pose = pose @ ob.inv4x4(g["Rt"][i])
        assert np.allclose(pose[:3, 3], g["trajectory"][i + 1], atol=1e-9)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", STEREO)
def test_gpu_reproduces_golden(name):
    g = load(name)
    H, W = g["lt1"].shape
    p = abi.default_params(W, H, num_disparities=int(g["num_disparities"]), max_features=4096)
    ctx = abi.Context(p)
    ctx.set_calib(g["P0"], g["P1"])
    ctx.set_images(g["lt1"], g["rt1"], g["lt2"], g["rt2"])
    ctx.compute_disparity()
    assert np.array_equal(ctx.disparity16(), g["disp16"])
    assert np.array_equal(ctx.tap(abi.TAP_DISP_PREMEDIAN), g["disp_premedian"])
    ctx.detect_fast()
    assert np.array_equal(ctx.features(), g["keypoints"])
    ctx.match_klt()
    legs, st = ctx.klt_legs(len(g["keypoints"]))
    assert np.array_equal(st, g["leg_status"])
    alive = g["leg_status"].min(axis=0) == 1
    assert np.abs(legs[:, alive] - g["legs"][:, alive]).max() <= 0.01
    t1, t2 = ctx.matches()
    assert t1.shape == g["match_t1"].shape
    assert np.abs(t1 - g["match_t1"]).max() <= 0.01 and np.abs(t2 - g["match_t2"]).max() <= 0.01
    assert np.array_equal(ctx.tap(abi.TAP_PYR_LEVEL, 0, 0, 1), g["pyr1"])
    assert np.array_equal(ctx.tap(abi.TAP_DERIV_LEVEL, 0, 0, 1), g["scharr1"])
    ctx.triangulate()
    obj, img2 = ctx.object_points()
    assert np.array_equal(obj, g["object_points"]) and np.array_equal(img2, g["object_img2"])
    Rt, info = ctx.estimate_motion()
    assert info[0].ok == 1 and info[0].best_inliers == int(g["pnp_inliers"]) and info[0].iters_run == int(g["pnp_iters"])
    want = g["Rt"]
    dR = Rt[0][:3, :3] @ want[:3, :3].T
    assert np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)) <= 1e-4
    assert np.abs(Rt[0][:3, 3] - want[:3, 3]).max() <= 1e-3
    ctx.close()


@pytest.mark.gpu
def test_gpu_hard_sgbm_both_tie_rules():
    g = load("sgbm_hard_200x60_d32")
    H, W = g["left"].shape
    for lanes, key in ((8, "disp16"), (0, "disp16_scalar_ties")):
        p = abi.default_params(W, H, num_disparities=32, wta_simd_lanes=lanes, max_features=1024)
        ctx = abi.Context(p)
        ctx.set_images(g["left"], g["right"], g["left"], g["right"])
        ctx.compute_disparity()
        assert np.array_equal(ctx.disparity16(), g[key]), lanes
        if lanes == 8:
            assert np.array_equal(ctx.tap(abi.TAP_DISP_PREMEDIAN), g["disp_premedian"])
        ctx.close()
