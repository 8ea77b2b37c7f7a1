"""The host-side pose solvers of the library (through the C ABI, no GPU needed) against things the CPU oracle did not
write: the pose that generated exact synthetic correspondences, and scipy's minimum of the same reprojection residual.
(The oracle's EPnP / iterative solver follow the same published algorithms as the library's, so agreement between those
two alone would not catch a shared mistake.)"""
import ctypes as C

import numpy as np
import pytest
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation

from visualodometry_amd import abi

K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1.0]])


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def lib_epnp(obj, img):
    obj = np.ascontiguousarray(obj, dtype=np.float32); img = np.ascontiguousarray(img, dtype=np.float32)
    r, t = np.zeros(3), np.zeros(3)
    assert abi.lib().vo_host_epnp(_p(obj), _p(img), len(obj), _p(K), _p(r), _p(t)) == 0
    return r, t


def lib_iterative(obj, img):
    obj = np.ascontiguousarray(obj, dtype=np.float64); img = np.ascontiguousarray(img, dtype=np.float64)
    r, t = np.zeros(3), np.zeros(3)
    rc = abi.lib().vo_host_pnp_iterative(_p(obj), _p(img), len(obj), _p(K), _p(r), _p(t))
    return rc, r, t


def lib_p3p(obj, img):
    obj = np.ascontiguousarray(obj, dtype=np.float32); img = np.ascontiguousarray(img, dtype=np.float32)
    r, t = np.zeros(3), np.zeros(3)
    rc = abi.lib().vo_host_p3p(_p(obj), _p(img), _p(K), _p(r), _p(t))
    return rc, r, t


def project(rvec, tvec, obj):
    Xc = Rotation.from_rotvec(rvec).apply(obj) + tvec
    return np.stack([K[0, 0] * Xc[:, 0] / Xc[:, 2] + K[0, 2], K[1, 1] * Xc[:, 1] / Xc[:, 2] + K[1, 2]], axis=1)


def scene(rng, n, planar=False):
    """n points 4..40 m in front of a camera that moved ~0.8 m with a small rotation (the shape of the path's input)"""
    obj = np.stack([rng.uniform(-8, 8, n), rng.uniform(-2, 2, n), rng.uniform(4, 40, n)], axis=1)
    if planar:   # a slanted wall
        nrm = np.array([0.3, 0.1, 1.0]); nrm /= np.linalg.norm(nrm)
        obj -= np.outer((obj - np.array([0, 0, 15.0])) @ nrm, nrm)
    rvec = rng.normal(0, 0.03, 3)
    tvec = np.array([rng.normal(0, 0.05), rng.normal(0, 0.02), -0.8 + rng.normal(0, 0.05)])
    return obj, rvec, tvec


def pose_close(r, t, r0, t0, rot_tol, t_tol):
    dR = Rotation.from_rotvec(r) * Rotation.from_rotvec(r0).inv()
    return dR.magnitude() <= rot_tol and np.abs(t - t0).max() <= t_tol


def scipy_minimum(obj, img, r0, t0):
    f = lambda x: (project(x[:3], x[3:], obj) - img).ravel()
    s = least_squares(f, np.concatenate([r0, t0]), method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=2000)
    return s.x[:3], s.x[3:], s


@pytest.mark.parametrize("seed", range(6))
def test_iterative_recovers_the_generating_pose_from_exact_correspondences(seed):
    rng = np.random.RandomState(seed)
    obj, rvec, tvec = scene(rng, 60)
    img = project(rvec, tvec, obj)
    rc, r, t = lib_iterative(obj, img)
    assert rc == 1 and pose_close(r, t, rvec, tvec, 1e-9, 1e-9)


@pytest.mark.parametrize("seed", range(6))
def test_iterative_reaches_scipys_minimum_of_the_reprojection_residual(seed):
    rng = np.random.RandomState(100 + seed)
    obj, rvec, tvec = scene(rng, 300)
    img = project(rvec, tvec, obj) + rng.normal(0, 0.7, (300, 2))   # pixel noise: the minimum is no longer the truth
    rc, r, t = lib_iterative(obj, img)
    assert rc == 1
    rs, ts, sol = scipy_minimum(obj, img, rvec, tvec)
    assert sol.success and pose_close(r, t, rs, ts, 1e-7, 1e-7)
    # and it is a minimum, not merely close to the start: the truth has a larger residual
    res = lambda a, b: np.sum((project(a, b, obj) - img) ** 2)
    assert res(r, t) <= res(rvec, tvec) and abs(res(r, t) - res(rs, ts)) <= 1e-9 * res(rs, ts)


@pytest.mark.parametrize("seed", range(4))
def test_planar_object_starts_from_the_homography_and_reaches_the_same_minimum(seed):
    rng = np.random.RandomState(200 + seed)
    obj, rvec, tvec = scene(rng, 80, planar=True)
    exact = project(rvec, tvec, obj)
    rc, r, t = lib_iterative(obj, exact)
    assert rc == 1 and pose_close(r, t, rvec, tvec, 1e-8, 1e-7)
    img = exact + rng.normal(0, 0.5, exact.shape)
    rc, r, t = lib_iterative(obj, img)
    rs, ts, sol = scipy_minimum(obj, img, rvec, tvec)
    assert rc == 1 and sol.success and pose_close(r, t, rs, ts, 1e-6, 1e-6)
    # five coplanar points are enough for the planar start (the 12-parameter DLT would need six)
    rc, r, t = lib_iterative(obj[:5], exact[:5])
    assert rc == 1 and pose_close(r, t, rvec, tvec, 1e-6, 1e-5)
    # ... but five points in general position are refused
    obj5, r5, t5 = scene(rng, 5)
    assert lib_iterative(obj5, project(r5, t5, obj5))[0] == -3


@pytest.mark.parametrize("n", [5, 50])
@pytest.mark.parametrize("seed", range(5))
def test_epnp_against_the_truth_and_against_the_minimum(n, seed):
    rng = np.random.RandomState(300 + 10 * n + seed)
    obj, rvec, tvec = scene(rng, n)
    obj = obj.astype(np.float32).astype(np.float64)
    exact = project(rvec, tvec, obj)
    # exact correspondences (up to the float storage of the points): EPnP returns the generating pose
    r, t = lib_epnp(obj, exact)
    assert pose_close(r, t, rvec, tvec, 2e-4, 5e-3)
    assert np.abs(project(r, t, obj) - exact).max() < 0.05
    if n < 6:
        return
    # noisy correspondences: within the RANSAC threshold (8 px) of the observations, and near the residual's minimum
    img = (exact + rng.normal(0, 0.5, exact.shape)).astype(np.float32).astype(np.float64)
    r, t = lib_epnp(obj, img)
    rs, ts, sol = scipy_minimum(obj, img, rvec, tvec)
    assert np.abs(project(r, t, obj) - img).max() < 8.0
    assert pose_close(r, t, rs, ts, 5e-3, 0.05)


@pytest.mark.parametrize("seed", range(10))
def test_p3p_on_four_points_returns_the_generating_pose(seed):
    rng = np.random.RandomState(400 + seed)
    obj, rvec, tvec = scene(rng, 4)
    obj = obj.astype(np.float32).astype(np.float64)
    img = project(rvec, tvec, obj)
    rc, r, t = lib_p3p(obj, img)
    assert rc == 1
    assert np.abs(project(r, t, obj) - img).max() < 1e-2          # float storage of the image points
    assert pose_close(r, t, rvec, tvec, 1e-4, 1e-2)


def test_p3p_three_collinear_points_have_no_solution():
    obj = np.array([[0, 0, 10.0], [1, 0, 10.0], [2, 0, 10.0], [0, 1, 12.0]])
    img = project(np.zeros(3), np.zeros(3), obj)
    rc, r, t = lib_p3p(obj, img)
    assert rc == 0


# ---- the oracle's own restatement of the two rare branches takes other numerical routes (resultant + iterative root finder
# + Horn's quaternion; isotropically normalised SVD homography): it must land on the same poses as the library and the truth
@pytest.mark.parametrize("seed", range(10))
def test_p3p_library_and_oracle_agree_by_different_routes(seed):
    from oracle import binding as ob
    rng = np.random.RandomState(500 + seed)
    obj, rvec, tvec = scene(rng, 4)
    obj = obj.astype(np.float32)
    img = (project(rvec, tvec, obj.astype(np.float64)) + rng.normal(0, 0.3, (4, 2))).astype(np.float32)   # noisy: the fourth point only ranks
    rc, r, t = lib_p3p(obj, img)
    ok, ro, to = ob.p3p(obj, img, K)
    assert rc == 1 and ok == 1
    assert pose_close(r, t, ro, to, 1e-7, 1e-6)
    # the three points that define the pose are reproduced exactly, whatever the fourth says
    assert np.abs(project(r, t, obj[:3].astype(np.float64)) - img[:3]).max() < 1e-3


def test_ransac_entry_with_four_and_with_planar_points():
    from oracle import binding as ob
    rng = np.random.RandomState(77)
    p = ob.default_params(1241, 376)
    obj, rvec, tvec = scene(rng, 4)
    obj = obj.astype(np.float32); img = project(rvec, tvec, obj.astype(np.float64)).astype(np.float32)
    ok, r, t, mask, info = ob.solve_pnp_ransac(obj, img, K, p)
    assert ok == 1 and mask.all() and pose_close(r, t, rvec, tvec, 1e-4, 1e-2)
    assert ob.solve_pnp_ransac(obj[:3], img[:3], K, p)[0] == 0          # fewer than four: the original asserts
    # a planar scene: RANSAC as usual, the final solve starts from the homography
    obj, rvec, tvec = scene(rng, 200, planar=True)
    obj = obj.astype(np.float32)
    img = (project(rvec, tvec, obj.astype(np.float64)) + rng.normal(0, 0.2, (200, 2))).astype(np.float32)
    img[::10] += 40.0                                                    # gross outliers
    ok, r, t, mask, info = ob.solve_pnp_ransac(obj, img, K, p)
    assert ok == 1 and info.final_from_refine == 1 and not mask[::10].any() and mask.sum() >= 170
    inl = mask.astype(bool)
    rs, ts, sol = scipy_minimum(obj[inl].astype(np.float64), img[inl].astype(np.float64), rvec, tvec)
    assert pose_close(r, t, rs, ts, 1e-6, 1e-6)
    rc, rl, tl = lib_iterative(obj[inl], img[inl])
    assert rc == 1 and pose_close(rl, tl, r, t, 1e-7, 1e-7)
