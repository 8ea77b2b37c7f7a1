"""ctypes loader for the CPU oracle (oracle/libvo_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg;
the shipped package never imports it.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Point2f(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class Point3f(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int),
        ("min_disparity", C.c_int), ("num_disparities", C.c_int), ("block_size", C.c_int),
        ("p1", C.c_int), ("p2", C.c_int), ("disp12_max_diff", C.c_int), ("pre_filter_cap", C.c_int),
        ("uniqueness_ratio", C.c_int), ("wta_simd_lanes", C.c_int),
        ("fast_threshold", C.c_int), ("fast_nms", C.c_int),
        ("lk_win", C.c_int), ("lk_max_level", C.c_int), ("lk_max_count", C.c_int),
        ("lk_epsilon", C.c_double), ("lk_min_eig", C.c_double),
        ("loop_threshold", C.c_int), ("depth_thresh", C.c_int),
        ("ransac_iters", C.c_int), ("ransac_reproj", C.c_float), ("ransac_conf", C.c_double),
        ("lk_accum", C.c_int),
    ]


class PnpInfo(C.Structure):
    _fields_ = [("ok", C.c_int), ("iters_run", C.c_int), ("best_inliers", C.c_int), ("final_from_refine", C.c_int)]


class StepInfo(C.Structure):
    _fields_ = [("n_fast", C.c_int), ("n_matched", C.c_int), ("n_obj", C.c_int), ("pnp", PnpInfo),
                ("ms_disparity", C.c_double), ("ms_fast", C.c_double), ("ms_klt", C.c_double), ("ms_motion", C.c_double)]


def build(force=False):
    so = os.path.join(_HERE, "libvo_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("fast.cpp", "sgbm.cpp", "lk.cpp", "pnp.cpp", "pnp_minimal.cpp", "pipeline.cpp", "vo_oracle.h", "linalg.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libvo_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_trajectory_rmse.restype = C.c_float
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def default_params(width, height, **kw):
    p = Params()
    lib().orc_default_params(C.byref(p), width, height)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


VAR_PATH_START_C, VAR_DISP12_ZERO, VAR_RAW_BORDER_PIXEL, VAR_LK_FLOAT_ACC, VAR_LK_NO_ERR_STATUS = 1, 2, 4, 8, 16


def set_variant(flags):
    """Sensitivity switches (vo_oracle.h): replaces assumptions about OpenCV's internals by their alternatives.  0 restores."""
    lib().orc_set_variant(C.c_uint(int(flags)))


def set_threads(n):
    lib().orc_set_threads(int(n))


def fast(img, threshold=20, nms=True, cap=None, with_score=False):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    cap = cap or (w * h // 4 + 16)
    out = np.zeros((cap, 2), dtype=np.float32)
    resp = np.zeros(cap, dtype=np.uint8)
    score = np.zeros((h, w), dtype=np.uint8) if with_score else None
    n = lib().orc_fast(_p(img, C.c_uint8), w, h, w, int(threshold), int(nms), _p(out, Point2f), _p(resp, C.c_uint8), cap,
                       _p(score, C.c_uint8) if with_score else None)
    n = min(n, cap)
    return (out[:n].copy(), resp[:n].copy(), score) if with_score else out[:n].copy()


def sgbm(left, right, params, with_premedian=False):
    left = np.ascontiguousarray(left, dtype=np.uint8)
    right = np.ascontiguousarray(right, dtype=np.uint8)
    h, w = left.shape
    d = np.zeros((h, w), dtype=np.int16)
    pre = np.zeros((h, w), dtype=np.int16)
    rc = lib().orc_sgbm3way(_p(left, C.c_uint8), _p(right, C.c_uint8), C.byref(params), _p(d, C.c_int16), _p(pre, C.c_int16))
    if rc != 0:
        raise ValueError("orc_sgbm3way: bad parameters")
    return (d, pre) if with_premedian else d


def sgbm_cost_row(left, right, params, stripe, y):
    left = np.ascontiguousarray(left, dtype=np.uint8)
    right = np.ascontiguousarray(right, dtype=np.uint8)
    w1 = params.width - params.num_disparities
    c = np.zeros((w1, params.num_disparities), dtype=np.int16)
    rc = lib().orc_sgbm_cost_row(_p(left, C.c_uint8), _p(right, C.c_uint8), C.byref(params), stripe, y, _p(c, C.c_int16))
    if rc != 0:
        raise ValueError("orc_sgbm_cost_row")
    return c


def disp_to_float(d16):
    d16 = np.ascontiguousarray(d16, dtype=np.int16)
    out = np.zeros(d16.shape, dtype=np.float32)
    lib().orc_disp_to_float(_p(d16, C.c_int16), d16.size, _p(out, C.c_float))
    return out


def depth_map(disp_f32, fx, baseline):
    """Returns (patched disparity, depth f64); the input array is patched in place like the reference does."""
    h, w = disp_f32.shape
    depth = np.zeros((h, w), dtype=np.float64)
    lib().orc_depth_map(_p(disp_f32, C.c_float), w, h, C.c_double(fx), C.c_double(baseline), _p(depth, C.c_double))
    return depth


def pyr_levels(w, h, win, max_level):
    return lib().orc_pyr_levels(w, h, win, max_level)


def pyrdown(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), dtype=np.uint8)
    lib().orc_pyrdown(_p(img, C.c_uint8), w, h, _p(out, C.c_uint8), out.shape[1], out.shape[0])
    return out


def scharr(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.zeros((h, w, 2), dtype=np.int16)
    lib().orc_scharr(_p(img, C.c_uint8), w, h, _p(out, C.c_int16))
    return out


def lk(prev, nxt, pts, params):
    prev = np.ascontiguousarray(prev, dtype=np.uint8)
    nxt = np.ascontiguousarray(nxt, dtype=np.uint8)
    pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 2)
    n = pts.shape[0]
    out = np.zeros((n, 2), dtype=np.float32)
    st = np.zeros(n, dtype=np.uint8)
    err = np.zeros(n, dtype=np.float32)
    h, w = prev.shape
    lib().orc_lk(_p(prev, C.c_uint8), _p(nxt, C.c_uint8), w, h, _p(pts, Point2f), n, _p(out, Point2f), _p(st, C.c_uint8),
                 _p(err, C.c_float), C.byref(params))
    return out, st, err


def match_klt(lt1, rt1, lt2, rt2, pts, params, with_legs=False):
    imgs = [np.ascontiguousarray(i, dtype=np.uint8) for i in (lt1, rt1, lt2, rt2)]
    pts1 = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 2).copy()
    n = pts1.shape[0]
    pts2 = np.zeros((max(n, 1), 2), dtype=np.float32)
    legs = np.zeros((4, max(n, 1), 2), dtype=np.float32)
    lst = np.zeros((4, max(n, 1)), dtype=np.uint8)
    m = lib().orc_match_klt(*[_p(i, C.c_uint8) for i in imgs], C.byref(params), _p(pts1, Point2f), n, _p(pts2, Point2f),
                            _p(legs, Point2f), _p(lst, C.c_uint8))
    if with_legs:
        return pts1[:m].copy(), pts2[:m].copy(), legs[:, :n], lst[:, :n]
    return pts1[:m].copy(), pts2[:m].copy()


def triangulate(t1, t2, depth, P0, depth_thresh=3000):
    t1 = np.ascontiguousarray(t1, dtype=np.float32).reshape(-1, 2)
    t2 = np.ascontiguousarray(t2, dtype=np.float32).reshape(-1, 2)
    depth = np.ascontiguousarray(depth, dtype=np.float64)
    h, w = depth.shape
    m = t1.shape[0]
    obj = np.zeros((max(m, 1), 3), dtype=np.float32)
    i1 = np.zeros((max(m, 1), 2), dtype=np.float32)
    i2 = np.zeros((max(m, 1), 2), dtype=np.float32)
    P0 = np.ascontiguousarray(P0, dtype=np.float64)
    k = lib().orc_triangulate(_p(t1, Point2f), _p(t2, Point2f), m, _p(depth, C.c_double), w, h, _p(P0, C.c_double),
                              int(depth_thresh), _p(obj, Point3f), _p(i1, Point2f), _p(i2, Point2f))
    return obj[:k].copy(), i1[:k].copy(), i2[:k].copy()


def rng_subsets(count, nsubsets, model_points=5):
    out = np.zeros((nsubsets, model_points), dtype=np.int32)
    lib().orc_rng_subsets(count, model_points, nsubsets, _p(out, C.c_int))
    return out


def _K(K):
    return np.ascontiguousarray(np.asarray(K, dtype=np.float64)[:3, :3])


def epnp(obj, img, K):
    obj = np.ascontiguousarray(obj, dtype=np.float32)
    img = np.ascontiguousarray(img, dtype=np.float32)
    K = _K(K)
    r = np.zeros(3)
    t = np.zeros(3)
    lib().orc_epnp(_p(obj, Point3f), _p(img, Point2f), obj.shape[0], _p(K, C.c_double), _p(r, C.c_double), _p(t, C.c_double))
    return r, t


def p3p(obj, img, K):
    obj = np.ascontiguousarray(obj, dtype=np.float32)
    img = np.ascontiguousarray(img, dtype=np.float32)
    K = _K(K)
    r = np.zeros(3)
    t = np.zeros(3)
    ok = lib().orc_p3p(_p(obj, Point3f), _p(img, Point2f), _p(K, C.c_double), _p(r, C.c_double), _p(t, C.c_double))
    return ok, r, t


def pnp_iterative(obj, img, K):
    obj = np.ascontiguousarray(obj, dtype=np.float64)
    img = np.ascontiguousarray(img, dtype=np.float64)
    K = _K(K)
    r = np.zeros(3)
    t = np.zeros(3)
    rc = lib().orc_pnp_iterative(_p(obj, C.c_double), _p(img, C.c_double), obj.shape[0], _p(K, C.c_double), _p(r, C.c_double),
                                 _p(t, C.c_double))
    return rc, r, t


def pnp_errors(obj, img, K, rvec, tvec):
    obj = np.ascontiguousarray(obj, dtype=np.float32)
    img = np.ascontiguousarray(img, dtype=np.float32)
    K = _K(K)
    rvec = np.ascontiguousarray(rvec, dtype=np.float64)
    tvec = np.ascontiguousarray(tvec, dtype=np.float64)
    err = np.zeros(obj.shape[0], dtype=np.float32)
    lib().orc_pnp_errors(_p(obj, Point3f), _p(img, Point2f), obj.shape[0], _p(K, C.c_double), _p(rvec, C.c_double),
                         _p(tvec, C.c_double), _p(err, C.c_float))
    return err


def solve_pnp_ransac(obj, img, K, params):
    obj = np.ascontiguousarray(obj, dtype=np.float32)
    img = np.ascontiguousarray(img, dtype=np.float32)
    K = _K(K)
    n = obj.shape[0]
    r = np.zeros(3)
    t = np.zeros(3)
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    info = PnpInfo()
    ok = lib().orc_solve_pnp_ransac(_p(obj, Point3f), _p(img, Point2f), n, _p(K, C.c_double), C.byref(params), _p(r, C.c_double),
                                    _p(t, C.c_double), _p(mask, C.c_uint8), C.byref(info))
    return ok, r, t, mask[:n], info


def rodrigues(r):
    r = np.ascontiguousarray(r, dtype=np.float64)
    R = np.zeros((3, 3))
    J = np.zeros((3, 9))
    lib().orc_rodrigues_vec2mat(_p(r, C.c_double), _p(R, C.c_double), _p(J, C.c_double))
    return R, J


def rodrigues_inv(R):
    R = np.ascontiguousarray(R, dtype=np.float64)
    r = np.zeros(3)
    lib().orc_rodrigues_mat2vec(_p(R, C.c_double), _p(r, C.c_double))
    return r


def inv4x4(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    out = np.zeros((4, 4))
    lib().orc_inv4x4(_p(A, C.c_double), _p(out, C.c_double))
    return out


def step(lt1, rt1, lt2, rt2, params, P0, P1, pose):
    """One iteration of the reference main loop.  pose (4x4 f64) is updated in place.  Returns (Rt, info)."""
    imgs = [np.ascontiguousarray(i, dtype=np.uint8) for i in (lt1, rt1, lt2, rt2)]
    P0 = np.ascontiguousarray(P0, dtype=np.float64)
    P1 = np.ascontiguousarray(P1, dtype=np.float64)
    Rt = np.zeros((4, 4))
    info = StepInfo()
    rc = lib().orc_step(*[_p(i, C.c_uint8) for i in imgs], C.byref(params), _p(P0, C.c_double), _p(P1, C.c_double),
                        _p(pose, C.c_double), _p(Rt, C.c_double), C.byref(info))
    if rc != 0:
        raise RuntimeError("orc_step failed")
    return Rt, info


def step_ex(lt1, rt1, lt2, rt2, params, P0, P1, pose):
    """orc_step_ex: one main-loop iteration that also returns the stage outputs.
    Returns (Rt, info, disp16, fast keypoints, matched T1 points, matched T2 points); pose is updated in place."""
    imgs = [np.ascontiguousarray(i, dtype=np.uint8) for i in (lt1, rt1, lt2, rt2)]
    h, w = imgs[0].shape
    P0 = np.ascontiguousarray(P0, dtype=np.float64)
    P1 = np.ascontiguousarray(P1, dtype=np.float64)
    Rt = np.zeros((4, 4))
    info = StepInfo()
    d16 = np.zeros((h, w), dtype=np.int16)
    cap = w * h // 4
    kp = np.zeros((cap, 2), dtype=np.float32)
    t1 = np.zeros((cap, 2), dtype=np.float32)
    t2 = np.zeros((cap, 2), dtype=np.float32)
    rc = lib().orc_step_ex(*[_p(i, C.c_uint8) for i in imgs], C.byref(params), _p(P0, C.c_double), _p(P1, C.c_double),
                           _p(pose, C.c_double), _p(Rt, C.c_double), C.byref(info), _p(d16, C.c_int16), _p(kp, Point2f),
                           _p(t1, Point2f), _p(t2, Point2f))
    if rc != 0:
        raise RuntimeError("orc_step_ex failed")
    return Rt, info, d16, kp[:info.n_fast].copy(), t1[:info.n_matched].copy(), t2[:info.n_matched].copy()


def trajectory_rmse(est, gt):
    est = np.ascontiguousarray(est, dtype=np.float64)
    gt = np.ascontiguousarray(gt, dtype=np.float64)
    return float(lib().orc_trajectory_rmse(_p(est, C.c_double), _p(gt, C.c_double), est.shape[0]))
