"""Known-answer tests that pin the CPU oracle to hand-checkable results.

The reference ships no tests or golden vectors for this path and OpenCV is not available, so these are the
anchors of the restatement: each case has an answer that follows from the definition of the algorithm.
"""
import math

import numpy as np
import pytest

from oracle import binding as ob


# ----------------------------------------------------------------------------- FAST
def test_fast_single_bright_dot():
    img = np.full((32, 40), 10, dtype=np.uint8)
    img[15, 20] = 200
    pts, resp, score = ob.fast(img, 20, True, with_score=True)
    # all 16 ring pixels are darker by 190 -> every arc minimum is 190 -> score = 190 - 1
    assert pts.tolist() == [[20.0, 15.0]] and resp.tolist() == [189]
    assert score[15, 20] == 189 and score.sum() == 189


def test_fast_needs_nine_contiguous():
    base = np.full((16, 16), 100, dtype=np.uint8)
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0),
            (-3, 1), (-2, 2), (-1, 3)]
    for n, expect in ((8, 0), (9, 1)):
        img = base.copy()
        for k in range(n):
            dx, dy = ring[(k + 5) % 16]  # the arc wraps around index 15 -> 0
            img[8 + dy, 8 + dx] = 160
        pts = ob.fast(img, 20, False)
        assert int(any((p == [8.0, 8.0]).all() for p in pts)) == expect


def test_fast_border_and_order():
    rng = np.random.RandomState(3)
    img = rng.randint(0, 256, size=(40, 50)).astype(np.uint8)
    pts = ob.fast(img, 20, True)
    assert len(pts) > 0
    assert pts[:, 0].min() >= 3 and pts[:, 0].max() <= 50 - 4 and pts[:, 1].min() >= 3 and pts[:, 1].max() <= 40 - 4
    keys = pts[:, 1] * 1000 + pts[:, 0]
    assert np.all(np.diff(keys) > 0)  # y ascending, then x ascending


def test_fast_nms_strictness():
    # two equal adjacent corners suppress each other (strict >)
    img = np.full((20, 24), 10, dtype=np.uint8)
    img[10, 10] = 200
    img[10, 11] = 200
    assert len(ob.fast(img, 20, True)) == 0
    assert len(ob.fast(img, 20, False)) >= 2


# ----------------------------------------------------------------------------- SGBM
def test_sgbm_constant_shift():
    rng = np.random.RandomState(5)
    W, H, shift = 200, 40, 7
    wide = rng.randint(0, 256, size=(H, W + 64)).astype(np.float64)
    for _ in range(2):
        wide[:, 1:-1] = (wide[:, :-2] + 2 * wide[:, 1:-1] + wide[:, 2:]) / 4
    wide = np.clip(np.rint(wide), 0, 255).astype(np.uint8)
    left = wide[:, 20:20 + W]
    right = wide[:, 20 + shift:20 + shift + W]  # right(x) = left(x + shift)  =>  disparity = shift
    p = ob.default_params(W, H, num_disparities=32)
    d, pre = ob.sgbm(left, right, p, with_premedian=True)
    assert np.all(d[:, :32] == -16)  # x < numDisparities is always invalid
    core = d[2:-2, 32 + 8:W - 8]
    assert (core == 16 * shift).mean() > 0.99
    assert set(np.unique(core)) <= {16 * shift, -16}


def test_sgbm_stripe_rows_not_divisible():
    # H = 50 -> stripes of 13 rows, overlap 6 + 2: exercises the stripe arithmetic; result must not depend on threads
    rng = np.random.RandomState(9)
    W, H = 120, 50
    left = rng.randint(0, 256, size=(H, W)).astype(np.uint8)
    right = np.roll(left, -3, axis=1)
    p = ob.default_params(W, H, num_disparities=32)
    ob.set_threads(1)
    a = ob.sgbm(left, right, p)
    ob.set_threads(4)
    b = ob.sgbm(left, right, p)
    ob.set_threads(1)
    assert np.array_equal(a, b)
    assert (a[:, 40:-6] == 48).mean() > 0.9


def test_sgbm_wta_tie_rules_differ_only_on_ties():
    rng = np.random.RandomState(11)
    W, H = 160, 32
    left = np.full((H, W), 77, dtype=np.uint8)          # flat image: every disparity ties
    right = left.copy()
    p8 = ob.default_params(W, H, num_disparities=32)
    p0 = ob.default_params(W, H, num_disparities=32, wta_simd_lanes=0)
    _, pre8 = ob.sgbm(left, right, p8, with_premedian=True)
    _, pre0 = ob.sgbm(left, right, p0, with_premedian=True)
    # with all costs equal, the path costs favour d = 0 .. identical under both rules on a flat image start
    assert pre0.shape == pre8.shape
    tex = rng.randint(0, 256, size=(H, W)).astype(np.uint8)
    a = ob.sgbm(tex, np.roll(tex, -2, axis=1), p8)
    b = ob.sgbm(tex, np.roll(tex, -2, axis=1), p0)
    assert (a == b).mean() > 0.98


def test_disp_float_and_depth_patch():
    d16 = np.array([[-16, 0, 16, 100, 1535]], dtype=np.int16)
    f = ob.disp_to_float(d16)
    assert f.tolist() == [[-1.0, 0.0, 1.0, 6.25, 95.9375]]
    fx, b = 718.856, 386.1448 / 718.856
    depth = ob.depth_map(f, fx, b)
    assert f[0, 0] == np.float32(0.1) and f[0, 1] == np.float32(0.1)  # patched in place
    assert depth[0, 0] == fx * b / float(np.float32(0.1)) and depth[0, 0] > 3000
    assert depth[0, 2] == fx * b / 1.0


# ----------------------------------------------------------------------------- pyramids / derivatives
def test_pyrdown_constant_and_ramp():
    img = np.full((21, 33), 93, dtype=np.uint8)
    assert np.all(ob.pyrdown(img) == 93) and ob.pyrdown(img).shape == (11, 17)
    ramp = np.tile((np.arange(40) * 4).astype(np.uint8), (12, 1))
    down = ob.pyrdown(ramp)
    assert np.all(down[:, 1:-1] == ramp[0, 2:-2:2][None, :])  # the symmetric kernel keeps a linear ramp


def test_pyramid_level_rule():
    assert ob.pyr_levels(1241, 376, 15, 3) == 3
    assert ob.pyr_levels(160, 64, 15, 3) == 2   # level 3 would be 20x8 <= 15
    assert ob.pyr_levels(320, 96, 15, 3) == 2   # 40x12
    assert ob.pyr_levels(3840, 2160, 21, 3) == 3


def test_scharr_ramp():
    ramp = np.tile((np.arange(30) * 3).astype(np.uint8), (10, 1))
    d = ob.scharr(ramp)
    assert np.all(d[:, 1:-1, 0] == 16 * 6) and np.all(d[:, :, 1] == 0)
    assert np.all(d[:, 0, 0] == 0)  # reflect-101: I(-1) == I(1)


# ----------------------------------------------------------------------------- LK
def smooth_image(W, H, dx=0.0, dy=0.0):
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    x = x - dx
    y = y - dy
    v = 128 + 50 * np.sin(x * 0.21) * np.cos(y * 0.17) + 40 * np.sin(x * 0.05 + y * 0.09) + 30 * np.cos(x * 0.13 - y * 0.23)
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def test_lk_recovers_translation():
    W, H = 320, 200
    a = smooth_image(W, H)
    b = smooth_image(W, H, 2.5, -1.25)
    pts = np.array([[80, 60], [160, 100], [240, 150], [100.5, 120.25]], dtype=np.float32)
    p = ob.default_params(W, H)
    out, st, err = ob.lk(a, b, pts, p)
    assert st.tolist() == [1, 1, 1, 1]
    assert np.abs(out - pts - np.array([2.5, -1.25])).max() < 0.05


def test_lk_status_rules():
    W, H = 320, 200
    a = smooth_image(W, H)
    flat = np.full((H, W), 100, dtype=np.uint8)
    p = ob.default_params(W, H)
    pts = np.array([[100, 100], [-40, 50], [1e5, 10]], dtype=np.float32)
    out, st, _ = ob.lk(flat, flat, pts, p)
    assert st.tolist() == [0, 0, 0]          # no texture -> min eigenvalue below threshold; outside -> level-0 bounds test
    assert out[0].tolist() == [100.0, 100.0]  # untouched guess
    out, st, _ = ob.lk(a, a, pts, p)
    assert st.tolist() == [1, 0, 0] and np.abs(out[0] - pts[0]).max() < 1e-3


def test_match_klt_quirks():
    W, H = 320, 200
    a = smooth_image(W, H)
    r = smooth_image(W, H, -6.0, 0.0)      # right view: shifted left
    a2 = smooth_image(W, H, 1.0, 0.5)
    r2 = smooth_image(W, H, -5.0, 0.5)
    p = ob.default_params(W, H)
    pts = np.array([[90, 70], [200, 120], [150.0, 100.0], [2.0, 3.0]], dtype=np.float32)
    t1, t2, legs, st = ob.match_klt(a, r, a2, r2, pts, p, with_legs=True)
    assert len(t1) == len(t2) >= 3
    # the output T1 points are the RE-TRACKED positions (sub-pixel), not the inputs
    assert np.abs(t1[:3] - pts[:3]).max() < 2.0 and not np.array_equal(t1[:3], pts[:3])
    assert np.abs(t2[:3] - (pts[:3] + [1.0, 0.5])).max() < 0.2


# ----------------------------------------------------------------------------- RNG / RANSAC bookkeeping
def py_rng_subsets(count, n):
    state = (1 << 64) - 1
    out = []
    for _ in range(n):
        idx = []
        while len(idx) < 5:
            state = ((state & 0xffffffff) * 4164903690 + (state >> 32)) & ((1 << 64) - 1)
            v = (state & 0xffffffff) % count
            if v not in idx:
                idx.append(v)
        out.append(idx)
    return out


def test_rng_stream():
    # first draw by hand: state = 2^64-1 -> 0xFFFFFFFF * 4164903690 + 0xFFFFFFFF
    s1 = 0xFFFFFFFF * 4164903690 + 0xFFFFFFFF
    assert (s1 & 0xffffffff) % 1000 == ob.rng_subsets(1000, 1)[0, 0]
    for count in (6, 37, 4000):
        assert ob.rng_subsets(count, 100).tolist() == py_rng_subsets(count, 100)


# ----------------------------------------------------------------------------- rotations / linear algebra
def test_rodrigues_known_and_roundtrip():
    R, J = ob.rodrigues([0, 0, math.pi / 2])
    assert np.allclose(R, [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-15)
    rng = np.random.RandomState(1)
    for _ in range(20):
        r = rng.normal(0, 1.0, 3)
        R, J = ob.rodrigues(r)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-13) and abs(np.linalg.det(R) - 1) < 1e-13
        assert np.allclose(ob.rodrigues_inv(R), r, atol=1e-10)
        for i in range(3):
            e = np.zeros(3); e[i] = 1e-6
            num = (ob.rodrigues(r + e)[0] - ob.rodrigues(r - e)[0]) / 2e-6
            assert np.allclose(J[i].reshape(3, 3), num, atol=1e-7)
    R0, J0 = ob.rodrigues([0, 0, 0])
    assert np.array_equal(R0, np.eye(3))


def test_inv4x4():
    rng = np.random.RandomState(2)
    for _ in range(10):
        R, _ = ob.rodrigues(rng.normal(0, 0.5, 3))
        A = np.eye(4); A[:3, :3] = R; A[:3, 3] = rng.normal(0, 2, 3)
        assert np.allclose(ob.inv4x4(A) @ A, np.eye(4), atol=1e-14)


# ----------------------------------------------------------------------------- PnP
K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1.0]])


def make_scene(n, seed, noise=0.0):
    rng = np.random.RandomState(seed)
    obj = np.stack([rng.uniform(-8, 8, n), rng.uniform(-2, 2, n), rng.uniform(5, 40, n)], axis=1).astype(np.float32)
    r = np.array([0.01, -0.03, 0.005]); t = np.array([0.05, -0.02, -0.8])
    R, _ = ob.rodrigues(r)
    cam = obj.astype(np.float64) @ R.T + t
    img = np.stack([cam[:, 0] / cam[:, 2] * K[0, 0] + K[0, 2], cam[:, 1] / cam[:, 2] * K[1, 1] + K[1, 2]], axis=1)
    img = (img + rng.normal(0, noise, img.shape)).astype(np.float32)
    return obj, img, r, t


def test_epnp_exact_pose():
    obj, img, r, t = make_scene(5, 21)
    rv, tv = ob.epnp(obj, img, K)
    assert np.allclose(rv, r, atol=2e-4) and np.allclose(tv, t, atol=5e-3)
    obj, img, r, t = make_scene(50, 22)
    rv, tv = ob.epnp(obj, img, K)
    assert np.allclose(rv, r, atol=1e-4) and np.allclose(tv, t, atol=2e-3)


def test_iterative_pose_and_error_codes():
    obj, img, r, t = make_scene(60, 23, noise=0.2)
    rc, rv, tv = ob.pnp_iterative(obj.astype(np.float64), img.astype(np.float64), K)
    assert rc == 1 and np.allclose(rv, r, atol=5e-4) and np.allclose(tv, t, atol=1e-2)
    rc, _, _ = ob.pnp_iterative(obj[:5].astype(np.float64), img[:5].astype(np.float64), K)
    assert rc == -3
    # a planar object (all points on the ground plane) starts from the plane's homography instead of being refused
    flat = obj.astype(np.float64).copy(); flat[:, 1] = 1.65
    R, _ = ob.rodrigues(r)
    cam = flat @ R.T + t
    fimg = np.stack([cam[:, 0] / cam[:, 2] * K[0, 0] + K[0, 2], cam[:, 1] / cam[:, 2] * K[1, 1] + K[1, 2]], axis=1)
    rc, rv, tv = ob.pnp_iterative(flat, fimg, K)
    assert rc == 1 and np.allclose(rv, r, atol=1e-7) and np.allclose(tv, t, atol=1e-6)


def test_pnp_errors_definition():
    obj, img, r, t = make_scene(30, 24, noise=1.0)
    err = ob.pnp_errors(obj, img, K, r, t)
    R, _ = ob.rodrigues(r)
    cam = obj.astype(np.float64) @ R.T + t
    proj = np.stack([cam[:, 0] / cam[:, 2] * K[0, 0] + K[0, 2], cam[:, 1] / cam[:, 2] * K[1, 1] + K[1, 2]], axis=1).astype(np.float32)
    want = ((img - proj).astype(np.float64) ** 2).sum(axis=1)
    assert np.allclose(err, want, rtol=1e-4, atol=1e-4)


def test_ransac_with_outliers():
    obj, img, r, t = make_scene(400, 25, noise=0.3)
    rng = np.random.RandomState(26)
    bad = rng.choice(400, 120, replace=False)
    img[bad] += rng.uniform(30, 200, (120, 2)).astype(np.float32)
    p = ob.default_params(1241, 376)
    ok, rv, tv, mask, info = ob.solve_pnp_ransac(obj, img, K, p)
    assert ok == 1 and info.final_from_refine == 1
    assert mask[bad].sum() <= 3 and mask.sum() >= 270
    assert np.allclose(rv, r, atol=1e-3) and np.allclose(tv, t, atol=2e-2)
    assert info.iters_run < 100  # the iteration budget shrinks once a 70 % consensus is found


def test_triangulate_truncation_and_threshold():
    P0 = np.array([[700.0, 0, 100, 0], [0, 700, 50, 0], [0, 0, 1, 0]])
    depth = np.full((100, 200), 5000.0)
    depth[10, 20] = 12.5
    depth[11, 21] = 3000.0     # == threshold stays (z > 3000 is the rejection test)
    t1 = np.array([[20.9, 10.9], [21.0, 11.0], [50.0, 50.0]], dtype=np.float32)
    t2 = t1 + 1
    obj, i1, i2 = ob.triangulate(t1, t2, depth, P0, 3000)
    assert len(obj) == 2
    z = np.float32(12.5)
    assert obj[0, 2] == z and obj[0, 0] == np.float32(float(z) * (float(t1[0, 0]) - 100.0) / 700.0)
    assert np.array_equal(i2, t2[:2])


def test_rmse_like_reference():
    est = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 1.0]])
    gt = np.array([[0, 0, 0], [1, 0, 1], [2, 0, 1.0]])
    assert ob.trajectory_rmse(est, gt) == pytest.approx(math.sqrt(1 / 3), rel=1e-6)
