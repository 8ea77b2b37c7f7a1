"""Context life cycle and threading contract through the C ABI: contexts are independent (own stream, own buffers), one
context is used by one thread at a time, and destroying a context gives its memory back."""
import threading

import numpy as np
import pytest

from visualodometry_amd import abi, synth

pytestmark = pytest.mark.gpu

W, H = 320, 96
PARAMS = dict(num_disparities=64, max_features=4096)


@pytest.fixture(scope="module")
def pairs():
    seq = synth.Sequence(seed=99, width=W, height=H)
    f = [seq.frame(i) for i in range(4)]
    return seq, f


def run_one(seq, f, i, batch=1):
    ctx = abi.Context(abi.default_params(W, H, max_batch=batch, **PARAMS))
    ctx.set_calib(seq.P0, seq.P1)
    ctx.set_images(f[i][0], f[i][1], f[i + 1][0], f[i + 1][1])
    Rt, info = ctx.step()
    d = ctx.disparity16(0).copy()
    ctx.close()
    return Rt[0], d, info[0].n_matched


def test_create_destroy_returns_device_memory(pairs):
    import torch
    seq, f = pairs
    run_one(seq, f, 0)          # first use loads the code object and creates the shared stream
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(12):
        run_one(seq, f, 0, batch=4)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert abs(free0 - free1) < 64 << 20, (free0, free1)   # nothing accumulates (a context of this size holds ~0.5 GB)


def test_contexts_on_concurrent_threads_are_independent(pairs):
    seq, f = pairs
    want = [run_one(seq, f, i) for i in range(3)]
    got = [None] * 3
    errs = []

    def work(i):
        try:
            ctx = abi.Context(abi.default_params(W, H, max_batch=1, **PARAMS))
            ctx.set_calib(seq.P0, seq.P1)
            for _ in range(5):   # repeated steps interleave with the other threads' steps
                ctx.set_images(f[i][0], f[i][1], f[i + 1][0], f[i + 1][1])
                Rt, info = ctx.step()
            got[i] = (Rt[0], ctx.disparity16(0).copy(), info[0].n_matched)
            ctx.close()
        except Exception as e:   # surfaced below
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for i in range(3):
        assert np.array_equal(got[i][0], want[i][0]) and np.array_equal(got[i][1], want[i][1]) and got[i][2] == want[i][2]


def test_use_after_close_is_an_error_not_a_crash(pairs):
    seq, f = pairs
    ctx = abi.Context(abi.default_params(W, H, max_batch=1, **PARAMS))
    ctx.close()
    ctx.close()   # idempotent
    with pytest.raises((abi.VoError, ValueError, RuntimeError)):
        ctx.step()


def test_padded_host_rows_give_the_same_results_as_packed_ones(pairs):
    """vo_set_images with a row stride larger than the width (re-pitched 2D upload) against tightly packed frames (one
    linear upload per image set)"""
    import ctypes as C
    seq, f = pairs
    imgs = [f[0][0], f[0][1], f[1][0], f[1][1]]
    ctx = abi.Context(abi.default_params(W, H, max_batch=1, **PARAMS))
    ctx.set_calib(seq.P0, seq.P1)
    ctx.set_images(*imgs)
    Rt_a, _ = ctx.step()
    d_a, k_a = ctx.disparity16(0).copy(), ctx.features(0).copy()
    pad = 7
    padded = [np.full((H, W + pad), 255, dtype=np.uint8) for _ in range(4)]
    for p, im in zip(padded, imgs):
        p[:, :W] = im
    ptr = [p.ctypes.data_as(C.c_void_p) for p in padded]
    rc = abi.lib().vo_set_images(ctx._h, 1, *ptr, C.c_size_t(W + pad), C.c_size_t((W + pad) * H))
    assert rc == 0
    Rt_b, _ = ctx.step()
    assert np.array_equal(Rt_a, Rt_b) and np.array_equal(ctx.disparity16(0), d_a) and np.array_equal(ctx.features(0), k_a)
    ctx.close()


def test_same_disparity_range_with_two_windows_in_one_process():
    """The cost kernel's dynamic LDS grows with block_size (ring slots held in LDS): a small-window context created first
    must not pin the kernel's LDS ceiling below what a larger window of the same disparity range needs."""
    from oracle import binding as ob
    w, h, D = 417, 57, 192
    seq = synth.Sequence(seed=5151, width=w, height=h)
    l, r = seq.frame(0)
    for bs in (3, 7, 13, 5):
        p = abi.default_params(w, h, max_batch=1, num_disparities=D, block_size=bs, max_features=1024)
        ctx = abi.Context(p)
        ctx.set_images(l, r, l, r)
        ctx.compute_disparity()
        op = ob.default_params(w, h, num_disparities=D, block_size=bs)
        assert np.array_equal(ctx.disparity16(0), ob.sgbm(l, r, op)), bs
        ctx.close()


def test_setters_are_refused_between_submit_and_finish(pairs):
    seq, f = pairs
    ctx = abi.Context(abi.default_params(W, H, max_batch=1, **PARAMS))
    ctx.set_calib(seq.P0, seq.P1)
    ctx.set_images(f[0][0], f[0][1], f[1][0], f[1][1])
    want, _ = ctx.step()
    ctx.step_submit()
    with pytest.raises(abi.VoError) as e:
        ctx.set_images(f[1][0], f[1][1], f[2][0], f[2][1])
    assert e.value.code == -5
    with pytest.raises(abi.VoError):
        ctx.set_features(np.zeros((4, 2), dtype=np.float32))
    Rt, _ = ctx.step_finish()
    assert np.array_equal(Rt, want)
    ctx.set_images(f[1][0], f[1][1], f[2][0], f[2][1])   # accepted again
    ctx.step()
    ctx.close()
