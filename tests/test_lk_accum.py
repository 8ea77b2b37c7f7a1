"""The tracker's accumulator modes (orc_params.lk_accum, oracle/lk.cpp): OpenCV's acctype is float in every build but 32-bit
ARM without NEON, and its vector blocks add partial sums in an order that depends on the instruction set.  The oracle
restates those orders (1 scan order, 4 NEON, 8 x86) next to the exact integer sums (0) that the GPU library computes.

Known-answer property: a float sum of integers is exact while every partial sum stays below 2^24, so on an image whose
gradients are small all four modes must give bit-identical tracks; on a full-contrast image they may differ, rarely and by
little.  GPU part: the library refuses the float modes instead of silently computing something else."""
import numpy as np
import pytest

from visualodometry_amd import abi, synth
from oracle import binding as ob

W, H = 320, 96


def _tracks(img0, img1, pts, mode):
    p = ob.default_params(W, H, lk_accum=mode)
    return ob.lk(img0, img1, pts, p)


def test_modes_agree_exactly_when_no_partial_sum_can_round():
    seq = synth.Sequence(seed=31, width=W, height=H, contrast=6.0)   # 20 times less contrast: products stay far below 2^24 / 225
    a, b = seq.frame(0)[0], seq.frame(1)[0]
    assert int(a.max()) - int(a.min()) < 24
    yy, xx = np.mgrid[12:H - 12:9, 12:W - 12:11]
    pts = np.stack([xx.ravel(), yy.ravel()], 1).astype(np.float32)
    base = _tracks(a, b, pts, 0)
    for mode in (1, 4, 8):
        got = _tracks(a, b, pts, mode)
        assert np.array_equal(got[1], base[1]), mode
        assert np.array_equal(got[0], base[0]), mode


def test_modes_differ_rarely_and_by_little_at_full_contrast():
    seq = synth.Sequence(seed=31, width=W, height=H)
    a, b = seq.frame(0)[0], seq.frame(1)[0]
    pts = ob.fast(a)
    assert len(pts) > 150
    base, st0, _ = _tracks(a, b, pts, 0)
    some_difference = False
    for mode in (1, 4, 8):
        got, st, _ = _tracks(a, b, pts, mode)
        both = (st0 == 1) & (st == 1)
        d = np.abs(got[both] - base[both]).max(axis=1)
        some_difference = some_difference or bool((d > 0).any())
        assert (d > 0.01).mean() < 0.02, (mode, float((d > 0.01).mean()))   # north-star tolerance: 0.01 px
        assert (st != st0).mean() < 0.01
    assert some_difference      # partial sums do reach 2^24 on corners of a full-contrast image


def test_default_is_integer_sums_on_both_sides():
    assert ob.default_params(W, H).lk_accum == 0 and abi.default_params(W, H).lk_accum == 0


@pytest.mark.gpu
def test_library_refuses_the_float_modes():
    for mode in (1, 4, 8):
        with pytest.raises(abi.VoError) as e:
            abi.Context(abi.default_params(W, H, lk_accum=mode))
        assert e.value.code == -8 and "lk_accum" in str(e.value)      # VO_ERR_UNSUPPORTED
