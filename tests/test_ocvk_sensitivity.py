"""tests/golden/ocvk_sensitivity.json records, for every assumption the oracle makes about OpenCV 4.5.4's internals, what
changes on KITTI-shape frame pairs when the assumption is replaced by its alternative.  The file must cover every switch
and must be reproducible: a sample of the switches is re-measured here on the first pair."""
import importlib.util
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_ocvk_sensitivity", os.path.join(HERE, "golden", "make_ocvk_sensitivity.py"))
mk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mk)


def recorded():
    return json.load(open(os.path.join(HERE, "golden", "ocvk_sensitivity.json")))


def test_every_assumption_has_a_measured_blast_radius():
    rec = recorded()
    assert set(rec["assumptions"]) == set(mk.ASSUMPTIONS)
    for name, v in rec["assumptions"].items():
        assert len(v["per_pair"]) == len(mk.PAIRS)
        assert v["beyond_parity_tolerance"] == (v["max_pose_rotation_change_rad"] > 1e-4 or v["max_pose_translation_change_m"] > 1e-3)
    # what the numbers say today: no single wrong guess moves a relative pose beyond the parity tolerance on these pairs,
    # the LR-check width is the one that changes the most disparity pixels, the LK accumulator type moves tracks by a
    # fraction of a pixel
    assert not any(v["beyond_parity_tolerance"] for v in rec["assumptions"].values())
    assert rec["assumptions"]["disp12_max_diff_zero"]["per_pair"][0]["disparity_pixels_changed"] > 10000
    assert rec["assumptions"]["path_start_state"]["per_pair"][0]["disparity_pixels_changed"] == 0


def test_recorded_numbers_are_reproducible():
    names = ("disp12_max_diff_zero", "lk_float_accumulators")
    got = mk.measure(names=names, pairs=mk.PAIRS[:1])
    rec = recorded()["assumptions"]
    for n in names:
        a, b = got[n]["per_pair"][0], rec[n]["per_pair"][0]
        for k in ("disparity_pixels_changed", "klt_status_changed", "matches", "object_points", "ransac_inliers", "ransac_iterations"):
            assert a[k] == b[k], (n, k)
        assert a["pose_rotation_change_rad"] == pytest.approx(b["pose_rotation_change_rad"], abs=1e-12)
        assert a["pose_translation_change_m"] == pytest.approx(b["pose_translation_change_m"], abs=1e-12)
