"""ctypes binding of the C ABI in include/vo_abi.h (visualodometry_amd/libvo_amd.so).

There is no CPU fallback: if the shared library is missing or no gfx950 device is present, the calls raise.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VO_AMD_LIB") or os.path.join(_HERE, "libvo_amd.so")  # override: A/B runs of kernel variants
_LIB = None

VO_OK = 0
STATUS_NAMES = {0: "VO_OK", -1: "VO_ERR_BAD_ARG", -2: "VO_ERR_BAD_PARAM", -3: "VO_ERR_HIP", -4: "VO_ERR_NO_DEVICE",
                -5: "VO_ERR_STATE", -6: "VO_ERR_CAPACITY", -7: "VO_ERR_NOT_ENOUGH_POINTS", -8: "VO_ERR_UNSUPPORTED"}

TAP_FAST_SCORE, TAP_DISP_PREMEDIAN, TAP_COST_ROW, TAP_PYR_LEVEL, TAP_DERIV_LEVEL = 1, 2, 3, 4, 5

# every symbol declared in include/vo_abi.h
EXPORTS = (
    "vo_default_params", "vo_create", "vo_destroy", "vo_last_error", "vo_create_error", "vo_stream", "vo_synchronize",
    "vo_set_calib", "vo_set_images", "vo_set_images_device", "vo_set_sequence", "vo_set_sequence_device",
    "vo_compute_disparity", "vo_get_disparity16",
    "vo_get_disparity32", "vo_compute_depth_map", "vo_get_depth", "vo_set_disparity32", "vo_set_depth", "vo_get_depth_cloud", "vo_detect_fast", "vo_get_features", "vo_set_features", "vo_detect_fast_noalloc",
    "vo_match_klt", "vo_get_matches", "vo_get_klt_legs", "vo_set_matches", "vo_triangulate", "vo_get_object_points", "vo_get_object_points_batch",
    "vo_pnp_score", "vo_estimate_motion", "vo_integrate_pose", "vo_step", "vo_step_submit", "vo_step_finish", "vo_read_tap", "vo_pyramid_info",
    "vo_set_timing", "vo_get_timing", "vo_stage_name",
    "vo_host_epnp", "vo_host_epnp5_fixed", "vo_host_pnp_iterative", "vo_host_p3p", "vo_host_rng_subsets", "vo_host_rodrigues",
)
NUM_STAGES = 16
# vo_cloud_point
CLOUD_DTYPE = np.dtype([("row", np.int32), ("col", np.int32), ("neg_z", np.float32), ("gray", np.uint8), ("pad", np.uint8, (3,))])


class Point2f(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class Point3f(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("max_batch", C.c_int), ("max_features", C.c_int),
        ("min_disparity", C.c_int), ("num_disparities", C.c_int), ("block_size", C.c_int),
        ("p1", C.c_int), ("p2", C.c_int), ("disp12_max_diff", C.c_int), ("pre_filter_cap", C.c_int),
        ("uniqueness_ratio", C.c_int), ("wta_simd_lanes", C.c_int),
        ("fast_threshold", C.c_int), ("fast_nms", C.c_int),
        ("lk_win", C.c_int), ("lk_max_level", C.c_int), ("lk_max_count", C.c_int),
        ("lk_epsilon", C.c_double), ("lk_min_eig", C.c_double),
        ("loop_threshold", C.c_int), ("depth_thresh", C.c_int),
        ("ransac_iters", C.c_int), ("ransac_reproj", C.c_float), ("ransac_conf", C.c_double),
        ("ransac_chunk", C.c_int), ("host_threads", C.c_int), ("lk_accum", C.c_int), ("pnp_device", C.c_int),
    ]


class FrameInfo(C.Structure):
    _fields_ = [("ok", C.c_int), ("iters_run", C.c_int), ("best_inliers", C.c_int), ("final_from_refine", C.c_int),
                ("n_features", C.c_int), ("n_matched", C.c_int), ("n_object_points", C.c_int)]


class VoError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("%s: %s" % (STATUS_NAMES.get(code, code), text))
        self.code = code


def lib():
    """Loads the shared library (raises OSError if it has not been built)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("%s not found: build it with `make -C visualodometry_amd/csrc` (python __graft_entry__.py)" % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64, and a process
        # that first loads the system copy (through this library) and later torch's finds no device from torch.  Loading
        # torch's copy first makes libvo_amd.so bind to the same runtime, so device pointers and streams can be shared.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        L.vo_last_error.restype = C.c_char_p
        L.vo_create_error.restype = C.c_char_p
        L.vo_stream.restype = C.c_void_p
        L.vo_stage_name.restype = C.c_char_p
        L.vo_destroy.restype = None
        for name in ("vo_set_images", "vo_set_images_device"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
        for name in ("vo_set_sequence", "vo_set_sequence_device"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
        L.vo_read_tap.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.vo_detect_fast_noalloc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.vo_pnp_score.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def default_params(width, height, **kw):
    p = Params()
    rc = lib().vo_default_params(C.byref(p), int(width), int(height))
    if rc != VO_OK:
        raise VoError(rc, "vo_default_params")
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One vo_ctx: a GPU stream plus all device buffers for batches of up to params.max_batch frame pairs."""

    def __init__(self, params, device=0):
        self.params = params
        self._h = C.c_void_p()
        self._L = lib()
        rc = self._L.vo_create(C.byref(params), int(device), C.byref(self._h))
        if rc != VO_OK:
            raise VoError(rc, self._L.vo_create_error().decode())
        self.W, self.H = params.width, params.height
        self.n = 0
        self._keep = None

    def close(self):
        if self._h:
            self._L.vo_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != VO_OK:
            raise VoError(rc, self._L.vo_last_error(self._h).decode())

    def stream(self):
        return self._L.vo_stream(self._h)

    def synchronize(self):
        self._check(self._L.vo_synchronize(self._h))

    def set_calib(self, P0, P1):
        P0 = np.ascontiguousarray(P0, dtype=np.float64).reshape(12)
        P1 = np.ascontiguousarray(P1, dtype=np.float64).reshape(12)
        self._check(self._L.vo_set_calib(self._h, _ptr(P0), _ptr(P1)))

    def set_images(self, lt1, rt1, lt2, rt2):
        """Host images: each H x W (n = 1) or n x H x W uint8 arrays."""
        arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in (lt1, rt1, lt2, rt2)]
        arrs = [a.reshape((1,) + a.shape) if a.ndim == 2 else a for a in arrs]
        n, h, w = arrs[0].shape
        if (h, w) != (self.H, self.W) or any(a.shape != arrs[0].shape for a in arrs):
            raise ValueError("image shape mismatch")
        self._check(self._L.vo_set_images(self._h, n, *[_ptr(a) for a in arrs], w, w * h))
        self.n = n

    def set_images_device(self, n, ptrs, stride, frame_stride, keep=None):
        """Device pointers (ints) of LT1, RT1, LT2, RT2; `keep` holds references that own the memory."""
        self._keep = keep
        self._check(self._L.vo_set_images_device(self._h, int(n), *[C.c_void_p(int(p)) for p in ptrs], int(stride), int(frame_stride)))
        self.n = int(n)

    def set_sequence(self, left, right):
        """Host images of n + 1 consecutive frames ((n+1) x H x W uint8 each): pair b is (frame b, frame b + 1)."""
        left = np.ascontiguousarray(left, dtype=np.uint8)
        right = np.ascontiguousarray(right, dtype=np.uint8)
        if left.ndim != 3 or left.shape != right.shape or left.shape[1:] != (self.H, self.W) or left.shape[0] < 2:
            raise ValueError("expected two (n+1) x H x W arrays")
        n = left.shape[0] - 1
        self._check(self._L.vo_set_sequence(self._h, n, _ptr(left), _ptr(right), self.W, self.W * self.H))
        self.n = n

    def set_sequence_device(self, n_pairs, left_ptr, right_ptr, stride, frame_stride, keep=None):
        self._keep = keep
        self._check(self._L.vo_set_sequence_device(self._h, int(n_pairs), C.c_void_p(int(left_ptr)), C.c_void_p(int(right_ptr)),
                                                   int(stride), int(frame_stride)))
        self.n = int(n_pairs)

    # --- VOClass::computeDisparity
    def compute_disparity(self):
        self._check(self._L.vo_compute_disparity(self._h))

    def disparity16(self, b=0):
        out = np.empty((self.H, self.W), dtype=np.int16)
        self._check(self._L.vo_get_disparity16(self._h, b, _ptr(out)))
        return out

    def disparity32(self, b=0):
        out = np.empty((self.H, self.W), dtype=np.float32)
        self._check(self._L.vo_get_disparity32(self._h, b, _ptr(out)))
        return out

    # --- VOClass::computeDepthMap
    def compute_depth_map(self):
        self._check(self._L.vo_compute_depth_map(self._h))

    def depth(self, b=0):
        out = np.empty((self.H, self.W), dtype=np.float64)
        self._check(self._L.vo_get_depth(self._h, b, _ptr(out)))
        return out

    def set_disparity32(self, disp, b=0):
        """Caller-supplied float disparity map for frame b (what computeDepthMap's argument holds)."""
        disp = np.ascontiguousarray(disp, dtype=np.float32)
        if disp.shape != (self.H, self.W):
            raise ValueError("disparity map shape mismatch")
        self._check(self._L.vo_set_disparity32(self._h, b, _ptr(disp)))

    def set_depth(self, depth, b=0):
        """Caller-supplied f64 depth map for frame b (what estimateMotion's depthMap argument holds)."""
        depth = np.ascontiguousarray(depth, dtype=np.float64)
        if depth.shape != (self.H, self.W):
            raise ValueError("depth map shape mismatch")
        self._check(self._L.vo_set_depth(self._h, b, _ptr(depth)))

    def depth_cloud(self, b=0):
        """The vertices VOClass::writeToPLY lists for frame b's depth map, compacted on the GPU: a structured array with
        fields row, col (int32), neg_z (float32), gray (uint8), in row-major order."""
        n = C.c_int()
        self._check(self._L.vo_get_depth_cloud(self._h, b, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=CLOUD_DTYPE)
        self._check(self._L.vo_get_depth_cloud(self._h, b, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out[:n.value]

    # --- VOClass::getFeaturesFAST
    def detect_fast(self):
        self._check(self._L.vo_detect_fast(self._h))

    def features(self, b=0):
        n = C.c_int()
        self._check(self._L.vo_get_features(self._h, b, None, 0, C.byref(n)))
        out = np.empty((max(n.value, 1), 2), dtype=np.float32)
        self._check(self._L.vo_get_features(self._h, b, _ptr(out), n.value, C.byref(n)))
        return out[:n.value]

    def set_features(self, pts, b=0):
        pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 2)
        self._check(self._L.vo_set_features(self._h, b, _ptr(pts), pts.shape[0]))

    # --- VOClass::matchFeatureKLT
    def match_klt(self):
        self._check(self._L.vo_match_klt(self._h))

    def matches(self, b=0):
        m = C.c_int()
        self._check(self._L.vo_get_matches(self._h, b, None, None, 0, C.byref(m)))
        t1 = np.empty((max(m.value, 1), 2), dtype=np.float32)
        t2 = np.empty((max(m.value, 1), 2), dtype=np.float32)
        self._check(self._L.vo_get_matches(self._h, b, _ptr(t1), _ptr(t2), m.value, C.byref(m)))
        return t1[:m.value], t2[:m.value]

    def klt_legs(self, n, b=0):
        pts = np.empty((4, max(n, 1), 2), dtype=np.float32)
        st = np.empty((4, max(n, 1)), dtype=np.uint8)
        if n > 0:
            self._check(self._L.vo_get_klt_legs(self._h, b, _ptr(pts), _ptr(st), n))
        return pts[:, :n], st[:, :n]

    def set_matches(self, t1, t2, b=0):
        t1 = np.ascontiguousarray(t1, dtype=np.float32).reshape(-1, 2)
        t2 = np.ascontiguousarray(t2, dtype=np.float32).reshape(-1, 2)
        self._check(self._L.vo_set_matches(self._h, b, _ptr(t1), _ptr(t2), t1.shape[0]))

    # --- VOClass::estimateMotion
    def triangulate(self):
        self._check(self._L.vo_triangulate(self._h))

    def object_points(self, b=0, k=None, with_image_points=True):
        """Triangulated points of frame b (and their T2 image points).  k: the count if the caller already knows it (the
        frame info of a step carries it), which saves the query; with_image_points=False fetches the 3-D points only."""
        if k is None:
            kk = C.c_int()
            self._check(self._L.vo_get_object_points(self._h, b, None, None, 0, C.byref(kk)))
            k = kk.value
        kk = C.c_int()
        obj = np.empty((max(k, 1), 3), dtype=np.float32)
        img = np.empty((max(k, 1), 2), dtype=np.float32) if with_image_points else None
        self._check(self._L.vo_get_object_points(self._h, b, _ptr(obj), _ptr(img) if with_image_points else None, k, C.byref(kk)))
        k = min(k, kk.value)
        return (obj[:k], img[:k]) if with_image_points else obj[:k]

    def object_points_batch(self, cap=None):
        """The triangulated points of every frame of the batch with one copy: a list of [k_b, 3] float32 arrays."""
        n = self.n
        counts = (C.c_int * n)()
        self._check(self._L.vo_get_object_points_batch(self._h, None, None, 0, counts))
        cap = max(1, max(counts) if cap is None else min(cap, max(counts)))
        obj = np.empty((n, cap, 3), dtype=np.float32)
        self._check(self._L.vo_get_object_points_batch(self._h, _ptr(obj), None, cap, counts))
        return [obj[b, :min(counts[b], cap)] for b in range(n)]

    def object_points_batch_counts(self):
        """the number of triangulated points of every frame of the batch (no copy, no synchronisation)"""
        counts = (C.c_int * self.n)()
        self._check(self._L.vo_get_object_points_batch(self._h, None, None, 0, counts))
        return list(counts)

    def object_points_batch_into(self, dst_ptr, cap):
        """The same copy into caller memory at address dst_ptr, host or device (on this context's GPU): [n][cap][3] float32,
        frame b's first min(k_b, cap) rows filled.  Returns the counts."""
        counts = (C.c_int * self.n)()
        self._check(self._L.vo_get_object_points_batch(self._h, C.c_void_p(dst_ptr), None, cap, counts))
        return list(counts)

    def pnp_score(self, Rt12, thresh2=64.0, b=0, with_masks=False, k=None):
        Rt12 = np.ascontiguousarray(Rt12, dtype=np.float64).reshape(-1, 12)
        H = Rt12.shape[0]
        counts = np.zeros(H, dtype=np.int32)
        masks = None
        if with_masks:
            masks = np.zeros((H, max(k, 1)), dtype=np.uint8)
        self._check(self._L.vo_pnp_score(self._h, b, _ptr(Rt12), H, C.c_float(thresh2), _ptr(counts), _ptr(masks) if with_masks else None))
        return (counts, masks[:, :k]) if with_masks else counts

    def estimate_motion(self):
        n = self.n
        Rt = np.zeros((n, 4, 4), dtype=np.float64)
        info = (FrameInfo * n)()
        self._check(self._L.vo_estimate_motion(self._h, _ptr(Rt), info))
        return Rt, list(info)

    def _check_step(self, rc, strict):
        # VO_ERR_NOT_ENOUGH_POINTS leaves every frame's outputs filled (the failed frames have ok = 0, identity [R|t])
        if rc == -7 and not strict:
            return
        self._check(rc)

    def step(self, strict=True):
        """One main-loop iteration for the n frame pairs that are set.  strict=False: frames with too few correspondences
        are reported through info[b].ok == 0 instead of raising, the other frames of the batch keep their results."""
        n = self.n
        Rt = np.zeros((n, 4, 4), dtype=np.float64)
        info = (FrameInfo * n)()
        self._check_step(self._L.vo_step(self._h, _ptr(Rt), info), strict)
        return Rt, list(info)

    def step_submit(self):
        self._check(self._L.vo_step_submit(self._h))

    def step_finish(self, strict=True):
        n = self.n
        Rt = np.zeros((n, 4, 4), dtype=np.float64)
        info = (FrameInfo * n)()
        self._check_step(self._L.vo_step_finish(self._h, _ptr(Rt), info), strict)
        return Rt, list(info)

    # --- per-stage timing
    def set_timing(self, enable=True):
        """enable = 2: timing with every step on one stream (a stage's interval then holds its kernels alone)."""
        self._check(self._L.vo_set_timing(self._h, int(enable)))

    def timing(self, reset=False):
        """{stage name: (accumulated ms, calls)}"""
        ms = (C.c_double * NUM_STAGES)()
        calls = (C.c_longlong * NUM_STAGES)()
        self._check(self._L.vo_get_timing(self._h, ms, calls, int(bool(reset))))
        out = {}
        for i in range(NUM_STAGES):
            name = self._L.vo_stage_name(i).decode()
            if name:
                out[name] = (ms[i], calls[i])
        return out

    # --- taps
    def pyramid_info(self, level=0):
        w, h, ml = C.c_int(), C.c_int(), C.c_int()
        self._check(self._L.vo_pyramid_info(self._h, level, C.byref(w), C.byref(h), C.byref(ml)))
        return w.value, h.value, ml.value

    def tap(self, which, b=0, arg0=0, arg1=0):
        if which == TAP_FAST_SCORE:
            out = np.empty((self.H, self.W), dtype=np.uint8)
        elif which == TAP_DISP_PREMEDIAN:
            out = np.empty((self.H, self.W), dtype=np.int16)
        elif which == TAP_COST_ROW:
            out = np.empty((self.W - self.params.num_disparities, self.params.num_disparities), dtype=np.int16)
        elif which == TAP_PYR_LEVEL:
            w, h, _ = self.pyramid_info(arg1)
            out = np.empty((h, w), dtype=np.uint8)
        elif which == TAP_DERIV_LEVEL:
            w, h, _ = self.pyramid_info(arg1)
            out = np.empty((h, w, 2), dtype=np.int16)
        else:
            raise ValueError(which)
        self._check(self._L.vo_read_tap(self._h, which, b, arg0, arg1, _ptr(out), out.nbytes))
        return out


def integrate_pose(pose, Rt):
    """pose <- pose @ inv(Rt) with the library's LU inverse (Source/VOClass.cpp:587)."""
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    Rt = np.ascontiguousarray(Rt, dtype=np.float64)
    rc = lib().vo_integrate_pose(_ptr(pose), _ptr(Rt))
    if rc != VO_OK:
        raise VoError(rc, "vo_integrate_pose")
    return pose
