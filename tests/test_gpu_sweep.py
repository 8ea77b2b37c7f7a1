"""Configuration sweep on the GPU against the CPU oracle: odd image sizes (widths that are not multiples of 4, heights that
do not divide into the 4 stripes), every disparity width the library accepts, several SGBM windows and LK windows.
Integer stages bit-exact, tracks and poses within the stated tolerances (LK 0.01 px, pose 1e-4 rad / 1e-3 m)."""
import numpy as np
import pytest

from visualodometry_amd import abi, synth
from oracle import binding as ob

pytestmark = pytest.mark.gpu

# (width, height, num_disparities, block_size, lk_win, wta_simd_lanes)
CONFIGS = [
    (97, 45, 32, 5, 9, 8),
    (201, 64, 64, 9, 15, 8),
    (333, 101, 96, 11, 21, 16),
    (259, 70, 128, 13, 15, 0),
    (322, 50, 160, 3, 31, 8),
    (417, 57, 192, 7, 15, 16),
    (390, 46, 224, 11, 3, 8),
    (450, 43, 256, 5, 15, 8),
    # ranges that are a multiple of 16 but not of 32
    (120, 49, 16, 7, 9, 8),
    (211, 53, 48, 11, 15, 8),
    (275, 47, 80, 5, 15, 16),
    (305, 58, 112, 9, 21, 0),
    (380, 45, 176, 3, 15, 8),
    (470, 41, 240, 11, 15, 8),
]


def oracle_params(p):
    names = [f[0] for f in ob.Params._fields_]
    return ob.default_params(p.width, p.height, **{k: getattr(p, k) for k in names if k not in ("width", "height") and hasattr(p, k)})


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: "%dx%d_d%d_b%d_w%d_t%d" % c)
def test_stages_against_oracle(cfg):
    W, H, D, bs, win, tie = cfg
    seq = synth.Sequence(seed=4000 + W, width=W, height=H)
    frames = [seq.frame(i) for i in range(2)]
    imgs = (frames[0][0], frames[0][1], frames[1][0], frames[1][1])
    p = abi.default_params(W, H, max_batch=1, num_disparities=D, block_size=bs, lk_win=win, wta_simd_lanes=tie, max_features=4096)
    ctx = abi.Context(p)
    ctx.set_calib(seq.P0, seq.P1)
    ctx.set_images(*imgs)
    op = oracle_params(p)

    # disparity: s16 map and its float form, bit-exact
    ctx.compute_disparity()
    want, want_pre = ob.sgbm(imgs[0], imgs[1], op, with_premedian=True)
    assert np.array_equal(ctx.tap(abi.TAP_DISP_PREMEDIAN), want_pre)
    assert np.array_equal(ctx.disparity16(0), want)
    assert np.array_equal(ctx.disparity32(0), ob.disp_to_float(want))
    # one aggregated cost row (width1 x D, whatever stride the library keeps it at)
    assert np.array_equal(ctx.tap(abi.TAP_COST_ROW, 0, 0, 3), ob.sgbm_cost_row(imgs[0], imgs[1], op, 0, 3))

    # FAST: same keypoints in the same order
    ctx.detect_fast()
    feats = ctx.features(0)
    want_f = ob.fast(imgs[0], 20, True)
    assert feats.shape == want_f.shape and np.array_equal(feats, want_f)

    # pyramids and derivatives of all four images, every level
    pts = want_f if len(want_f) else np.array([[W / 2.0, H / 2.0]], dtype=np.float32)
    ctx.set_features(pts)
    ctx.match_klt()
    _, _, max_level = ctx.pyramid_info(0)
    assert max_level == ob.pyr_levels(W, H, p.lk_win, p.lk_max_level)
    for slot, img in enumerate(imgs):
        cur = img
        for level in range(max_level + 1):
            if level > 0:
                cur = ob.pyrdown(cur)
            assert np.array_equal(ctx.tap(abi.TAP_PYR_LEVEL, 0, slot, level), cur), (slot, level)
            assert np.array_equal(ctx.tap(abi.TAP_DERIV_LEVEL, 0, slot, level), ob.scharr(cur)), (slot, level)

    # circular tracking: statuses equal, points within 0.01 px
    w1, w2, legs, lst = ob.match_klt(*imgs, pts, op, with_legs=True)
    g_legs, g_st = ctx.klt_legs(len(pts))
    assert np.array_equal(g_st, lst)
    for leg in range(4):
        ok = lst[:leg + 1].min(axis=0) == 1
        d = np.abs(g_legs[leg][ok] - legs[leg][ok])
        assert d.size == 0 or d.max() <= 0.01, (leg, d.max())
    g1, g2 = ctx.matches(0)
    assert g1.shape == w1.shape
    if len(w1):
        assert np.abs(g1 - w1).max() <= 0.01 and np.abs(g2 - w2).max() <= 0.01

    # the whole step: same counts, same RANSAC trace, same relative pose (or the same failure) as the oracle
    Rt, info = ctx.step()
    want, oinfo = ob.step(*imgs, op, seq.P0, seq.P1, np.eye(4))
    assert info[0].n_features == oinfo.n_fast and info[0].n_matched == oinfo.n_matched and info[0].n_object_points == oinfo.n_obj
    assert info[0].ok == oinfo.pnp.ok
    if oinfo.pnp.ok:
        assert info[0].best_inliers == oinfo.pnp.best_inliers and info[0].iters_run == oinfo.pnp.iters_run
        assert np.abs(Rt[0][:3, 3] - want[:3, 3]).max() <= 1e-3
        assert np.abs(Rt[0][:3, :3] - want[:3, :3]).max() <= 1e-4
    ctx.close()


def test_batch_with_a_textureless_pair_in_the_middle():
    """frames of a batch are independent: a pair without any corner (no features, no pose) between two ordinary pairs
    leaves their results untouched and is itself reported like the oracle reports it"""
    W, H = 320, 96
    seq = synth.Sequence(seed=1234, width=W, height=H)
    f = [seq.frame(i) for i in range(3)]
    flat = np.full((H, W), 77, dtype=np.uint8)
    lt1 = np.stack([f[0][0], flat, f[1][0]]); rt1 = np.stack([f[0][1], flat, f[1][1]])
    lt2 = np.stack([f[1][0], flat, f[2][0]]); rt2 = np.stack([f[1][1], flat, f[2][1]])
    p = abi.default_params(W, H, max_batch=3, num_disparities=64, max_features=4096)
    ctx = abi.Context(p)
    ctx.set_calib(seq.P0, seq.P1)
    ctx.set_images(lt1, rt1, lt2, rt2)
    with pytest.raises(abi.VoError) as e:
        ctx.step()
    assert e.value.code == -7
    Rt, info = ctx.step(strict=False)
    assert np.array_equal(Rt[1], np.eye(4))
    op = oracle_params(p)
    for b in range(3):
        want, oinfo = ob.step(lt1[b], rt1[b], lt2[b], rt2[b], op, seq.P0, seq.P1, np.eye(4))
        assert info[b].n_features == oinfo.n_fast and info[b].n_matched == oinfo.n_matched and info[b].n_object_points == oinfo.n_obj
        assert bool(info[b].ok) == bool(oinfo.pnp.ok), b
        if oinfo.pnp.ok:
            assert np.abs(Rt[b] - want).max() <= 1e-3
        assert np.array_equal(ctx.disparity16(b), ob.sgbm(lt1[b], rt1[b], op))
    assert info[1].n_features == 0 and not info[1].ok
    # the ordinary pairs give what they give alone
    one = abi.Context(abi.default_params(W, H, max_batch=1, num_disparities=64, max_features=4096))
    one.set_calib(seq.P0, seq.P1)
    one.set_images(lt1[2], rt1[2], lt2[2], rt2[2])
    Rt1, _ = one.step()
    assert np.array_equal(Rt1[0], Rt[2])
    one.close(); ctx.close()


def test_odd_batch_equals_pairs_processed_alone():
    """11 frame pairs in one batch (not a multiple of anything a kernel tiles by: the tracker deals feature groups over 8 XCDs,
    the disparity kernels pack 4 rows per wavefront, the LR check 4 rows per block), pairs with different corner counts:
    every pair's matches and pose are bit-identical to the same pair processed alone, for both forms of input"""
    W, H, B = 283, 78, 11
    seq = synth.Sequence(seed=777, width=W, height=H)
    f = [seq.frame(i) for i in range(B + 1)]
    left = np.stack([x[0] for x in f]); right = np.stack([x[1] for x in f])
    # different corner counts per pair: blank out growing parts of some left images (both views of the stereo pair stay consistent enough)
    for i in (2, 5, 9):
        left[i, : 8 * i] = 60; right[i, : 8 * i] = 60
    kw = dict(num_disparities=48, block_size=7, max_features=2048)
    alone = []
    one = abi.Context(abi.default_params(W, H, max_batch=1, **kw))
    one.set_calib(seq.P0, seq.P1)
    for b in range(B):
        one.set_images(left[b], right[b], left[b + 1], right[b + 1])
        Rt, info = one.step(strict=False)
        alone.append((Rt[0].copy(), info[0].n_features, info[0].n_matched, one.disparity16(0).copy(), [m.copy() for m in one.matches(0)]))
    one.close()
    assert len({a[1] for a in alone}) > 3
    ctx = abi.Context(abi.default_params(W, H, max_batch=B, **kw))
    ctx.set_calib(seq.P0, seq.P1)
    for mode in ("pairs", "sequence"):
        if mode == "pairs":
            ctx.set_images(left[:B], right[:B], left[1:], right[1:])
        else:
            ctx.set_sequence(left, right)
        Rt, info = ctx.step(strict=False)
        for b in range(B):
            want_Rt, nf, nm, disp, m = alone[b]
            assert info[b].n_features == nf and info[b].n_matched == nm, (mode, b)
            assert np.array_equal(ctx.disparity16(b), disp), (mode, b)
            g1, g2 = ctx.matches(b)
            assert np.array_equal(g1, m[0]) and np.array_equal(g2, m[1]), (mode, b)
            assert np.array_equal(Rt[b], want_Rt), (mode, b)
    ctx.close()
