"""Regenerates tests/golden/*.npz: small seeded inputs and the outputs of the CPU oracle on them.

The reference holds no fixtures for this path (no tests, and its only numeric artefact needs the KITTI images),
so these vectors are produced by this repository's own oracle.  They freeze the oracle's behaviour (a change in
the restatement shows up as a diff here) and give the GPU tests inputs/outputs that do not depend on the oracle
being rebuilt identically.  Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from visualodometry_amd import synth  # noqa: E402
from oracle import binding as ob  # noqa: E402


def stereo_case(name, W, H, D, seed):
    seq = synth.Sequence(seed=seed, width=W, height=H)
    f0, f1 = seq.frame(0), seq.frame(1)
    p = ob.default_params(W, H, num_disparities=D)
    d16, pre = ob.sgbm(f0[0], f0[1], p, with_premedian=True)
    kp, resp, _ = ob.fast(f0[0], 20, True, with_score=True)
    t1, t2, legs, st = ob.match_klt(f0[0], f0[1], f1[0], f1[1], kp, p, with_legs=True)
    disp = ob.disp_to_float(d16)
    depth = ob.depth_map(disp, seq.P0[0, 0], seq.P1[0, 3] / (-seq.P0[0, 0]))
    obj, i1, i2 = ob.triangulate(t1, t2, depth, seq.P0)
    ok, rvec, tvec, mask, info = ob.solve_pnp_ransac(obj, i2, seq.P0, p)
    pose = np.eye(4)
    Rt, sinfo = ob.step(f0[0], f0[1], f1[0], f1[1], p, seq.P0, seq.P1, pose)
    lv1 = ob.pyrdown(f0[0])
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        lt1=f0[0], rt1=f0[1], lt2=f1[0], rt2=f1[1], P0=seq.P0, P1=seq.P1, num_disparities=D,
                        disp16=d16, disp_premedian=pre, keypoints=kp, responses=resp, legs=legs, leg_status=st,
                        match_t1=t1, match_t2=t2, object_points=obj, object_img2=i2, rvec=rvec, tvec=tvec,
                        inlier_mask=mask, pnp_iters=info.iters_run, pnp_inliers=info.best_inliers, Rt=Rt,
                        pyr1=lv1, scharr1=ob.scharr(lv1))
    print(name, "features", len(kp), "matches", len(t1), "object points", len(obj), "inliers", info.best_inliers,
          "invalid px", int((d16 < 0).sum()))


if __name__ == "__main__":
    stereo_case("stereo_320x96_d64", 320, 96, 64, 77)
    # height not divisible by 4 and a disparity range that leaves many left-right failures
    stereo_case("stereo_250x75_d32", 250, 75, 32, 78)
    # disparity-only case built to hit the rare branches: a noise block in the right view (left-right check
    # failures), a textureless band (exact cost ties in the winner-take-all) and saturated pixels
    rng = np.random.RandomState(79)
    W, H, D = 200, 60, 32
    left = synth.random_image(79, W + 8, H)[:, :W].copy()
    right = np.roll(left, -5, axis=1).copy()
    right[8:28, 60:110] = rng.randint(0, 256, size=(20, 50))
    # value 15 equals the prefilter's border constant: with the whole last stripe (and its warm-up rows) flat,
    # all disparities tie exactly there
    left[30:, :] = 15
    right[30:, :] = 15
    left[:6, :] = 255
    right[:6, :] = 255
    p = ob.default_params(W, H, num_disparities=D)
    d16, pre = ob.sgbm(left, right, p, with_premedian=True)
    p0 = ob.default_params(W, H, num_disparities=D, wta_simd_lanes=0)
    d16_scalar = ob.sgbm(left, right, p0)
    np.savez_compressed(os.path.join(HERE, "sgbm_hard_200x60_d32.npz"), left=left, right=right, num_disparities=D,
                        disp16=d16, disp_premedian=pre, disp16_scalar_ties=d16_scalar)
    print("sgbm_hard: invalid beyond the left band", int((d16[:, D:] < 0).sum()), "tie-rule differences", int((d16 != d16_scalar).sum()))
    rng_sets = {str(c): ob.rng_subsets(c, 100) for c in (6, 100, 4000)}
    np.savez_compressed(os.path.join(HERE, "rng_subsets.npz"), **rng_sets)
