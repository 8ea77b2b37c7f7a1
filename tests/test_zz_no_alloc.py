"""The allocation-free entry point (vo_detect_fast_noalloc) against vo_detect_fast + vo_get_features, through the C ABI:
same corners in the same order, on two calls in a row, the second on a smaller image (the shared scratch then still holds
the first call's scores beyond the second image's extent)."""
import ctypes as C

import numpy as np
import pytest

from visualodometry_amd import abi, synth

pytestmark = pytest.mark.gpu


def reference_corners(img, thr):
    h, w = img.shape
    ctx = abi.Context(abi.default_params(w, h, max_batch=1, num_disparities=32, max_features=65536, fast_threshold=thr))
    ctx.set_images(img, img, img, img)
    ctx.detect_fast()
    out = ctx.features(0).copy()
    ctx.close()
    return out


def noalloc_corners(img, thr, cap, stream=None):
    import torch
    h, w = img.shape
    d_img = torch.from_numpy(np.ascontiguousarray(img)).cuda()
    d_out = torch.full((cap, 2), -1.0, dtype=torch.float32, device="cuda")
    d_n = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    rc = abi.lib().vo_detect_fast_noalloc(d_img.data_ptr(), w, h, w, thr, 1, d_out.data_ptr(), cap, d_n.data_ptr(),
                                          C.c_void_p(stream.cuda_stream) if stream is not None else None)
    assert rc == 0, abi.lib().vo_create_error().decode()
    torch.cuda.synchronize()
    n = int(d_n.item())
    return d_out.cpu().numpy()[:n], n


def test_noalloc_fast_matches_the_context_entry_point_twice_in_a_row():
    big = synth.Sequence(seed=31, width=1241, height=376).frame(0)[0]
    small = synth.Sequence(seed=32, width=333, height=101).frame(0)[0]
    for img, thr in ((big, 20), (small, 12), (big, 35)):
        want = reference_corners(img, thr)
        assert len(want) > 100
        got, n = noalloc_corners(img, thr, 65536)
        assert n == len(want) and np.array_equal(got, want)
    # on a stream of the caller's instead of the NULL stream
    import torch
    side = torch.cuda.Stream()
    want = reference_corners(small, 12)
    got, n = noalloc_corners(small, 12, 65536, stream=side)
    assert n == len(want) and np.array_equal(got, want)
    # capacity smaller than the corner count: the first `cap` corners in row-major order, count clamped
    want = reference_corners(big, 20)
    got, n = noalloc_corners(big, 20, 500)
    assert n == 500 and np.array_equal(got, want[:500])


def test_noalloc_fast_rejects_what_does_not_fit():
    import torch
    lib = abi.lib()
    d = torch.zeros(16, dtype=torch.uint8, device="cuda")
    o = torch.zeros((4, 2), dtype=torch.float32, device="cuda")
    n = torch.zeros(1, dtype=torch.int32, device="cuda")
    rc = lib.vo_detect_fast_noalloc(d.data_ptr(), 4096, 2048, 4096, 20, 1, o.data_ptr(), 4, n.data_ptr(), None)   # 8 Mi pixels
    assert rc == -6 and b"exceeds" in lib.vo_create_error()
    assert lib.vo_detect_fast_noalloc(None, 64, 64, 64, 20, 1, o.data_ptr(), 4, n.data_ptr(), None) == -1
    assert lib.vo_detect_fast_noalloc(d.data_ptr(), 64, 64, 32, 20, 1, o.data_ptr(), 4, n.data_ptr(), None) == -1
