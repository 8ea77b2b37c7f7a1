#!/usr/bin/env python3
"""Blast radius of every assumption the CPU oracle makes about OpenCV 4.5.4's internals.

The reference's arithmetic lives in OpenCV 4.5.4, which is neither vendored nor installed here, so a handful of details of
the restatement rest on knowledge of its sources rather than on outputs.  This script flips each of them to its plausible
alternative, one at a time, in the oracle, runs the reference's main-loop step (Source/main.cpp:36-60) on KITTI-shape
synthetic frame pairs and records what changes: disparity pixels, circular matches, inliers, and the relative pose.  It
pins nothing; it says which guesses could move a trajectory beyond the 1e-4 rad / 1e-3 m parity tolerance if wrong.

    python tests/golden/make_ocvk_sensitivity.py            # rewrites tests/golden/ocvk_sensitivity.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob          # noqa: E402
from visualodometry_amd import synth      # noqa: E402

W, H = 1241, 376
PAIRS = ((1000, 0), (1003, 7))            # (sequence seed, first frame of the pair)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ocvk_sensitivity.json")

# name -> (variant bits, parameter overrides, what is assumed, what the alternative is)
ASSUMPTIONS = {
    "wta_tie_rule_scalar": (0, {"wta_simd_lanes": 0}, "winner-take-all ties follow the 8-lane (128-bit SIMD) loop: per lane the last minimum, then the smallest",
                            "scalar loop: the first minimum"),
    "wta_tie_rule_16_lanes": (0, {"wta_simd_lanes": 16}, "8 SIMD lanes", "16 lanes (256-bit build)"),
    "path_start_state": (ob.VAR_PATH_START_C, {}, "a path starts from zero costs with zero minimum: first pixel L = C - P2", "first pixel L = C"),
    "disp12_max_diff_zero": (ob.VAR_DISP12_ZERO, {}, "disp12MaxDiff = 0 is taken as 1 by the 3-way code", "0 stays 0"),
    "raw_channel_border": (ob.VAR_RAW_BORDER_PIXEL, {}, "the raw-intensity channel of the pixel cost holds the table's zero entry (15) in columns 0 and W-1",
                           "it holds the pixel"),
    "lk_float_accumulators": (ob.VAR_LK_FLOAT_ACC, {}, "LK structure tensor and mismatch vector are exact integer sums (lk_accum = 0; what the GPU library computes)",
                              "float sums in plain scan order (lk_accum = 1)"),
    "lk_float_accumulators_neon_order": (0, {"lk_accum": 4}, "exact integer sums (lk_accum = 0)",
                                         "float sums in the order of OpenCV's NEON block (lk_accum = 4: arm64, the build of the reference's author; "
                                         "acctype is float in every build but 32-bit ARM without NEON)"),
    "lk_float_accumulators_sse_order": (0, {"lk_accum": 8}, "exact integer sums (lk_accum = 0)",
                                        "float sums in the order of OpenCV's 128-bit x86 block (lk_accum = 8)"),
    "lk_error_pass_status": (ob.VAR_LK_NO_ERR_STATUS, {}, "after the iterations the final window position is tested again and can clear the status",
                             "no such test"),
}


def rot_angle(Ra, Rb):
    c = (np.trace(Ra.T @ Rb) - 1) / 2
    return float(np.arccos(min(1.0, max(-1.0, c))))


def run(seed, first, variant, overrides):
    seq = synth.Sequence(seed=seed, width=W, height=H)
    (l1, r1), (l2, r2) = seq.frame(first), seq.frame(first + 1)
    p = ob.default_params(W, H, **overrides)
    ob.set_variant(variant)
    try:
        d16 = ob.sgbm(l1, r1, p)
        feats = ob.fast(l1)
        m1, m2, legs, lst = ob.match_klt(l1, r1, l2, r2, feats, p, with_legs=True)
        keep = np.zeros(len(feats), dtype=bool)   # which FAST features closed the circle (loop closure is applied after the legs)
        Rt, info = ob.step(l1, r1, l2, r2, p, seq.P0, seq.P1, np.eye(4))
    finally:
        ob.set_variant(0)
    return {"d16": d16, "legs": legs, "lst": lst, "m1": m1, "Rt": Rt, "n_matched": info.n_matched, "n_obj": info.n_obj,
            "inliers": info.pnp.best_inliers, "iters": info.pnp.iters_run, "ok": info.pnp.ok}


def compare(base, var):
    valid = base["d16"] >= 0
    changed = base["d16"] != var["d16"]
    both = (base["lst"].min(axis=0) == 1) & (var["lst"].min(axis=0) == 1)
    leg_d = np.abs(base["legs"][:, both] - var["legs"][:, both]).max() if both.any() else 0.0
    per_track = np.abs(base["legs"][:, both] - var["legs"][:, both]).max(axis=(0, 2)) if both.any() else np.zeros(0)
    return {
        "disparity_pixels_changed": int(changed.sum()),
        "disparity_pixels_changed_frac_of_valid": float(changed.sum() / max(int(valid.sum()), 1)),
        "disparity_max_abs_change_16ths": int(np.abs(base["d16"].astype(np.int32) - var["d16"].astype(np.int32)).max()),
        "klt_status_changed": int((base["lst"] != var["lst"]).any(axis=0).sum()),
        "klt_max_track_change_px": float(leg_d),
        "klt_tracks_compared": int(both.sum()),
        "klt_tracks_moved_beyond_0.01px": int((per_track > 0.01).sum()),
        "klt_tracks_moved_beyond_1px": int((per_track > 1.0).sum()),
        "matches": [int(base["n_matched"]), int(var["n_matched"])],
        "object_points": [int(base["n_obj"]), int(var["n_obj"])],
        "ransac_inliers": [int(base["inliers"]), int(var["inliers"])],
        "ransac_iterations": [int(base["iters"]), int(var["iters"])],
        "pose_rotation_change_rad": rot_angle(base["Rt"][:3, :3], var["Rt"][:3, :3]),
        "pose_translation_change_m": float(np.abs(base["Rt"][:3, 3] - var["Rt"][:3, 3]).max()),
    }


def measure(names=None, pairs=PAIRS):
    ob.set_threads(min(8, os.cpu_count() or 1))
    out = {}
    bases = [run(seed, first, 0, {}) for seed, first in pairs]
    for name, (variant, overrides, assumed, alternative) in ASSUMPTIONS.items():
        if names and name not in names:
            continue
        per = [compare(b, run(seed, first, variant, overrides)) for b, (seed, first) in zip(bases, pairs)]
        rot = max(p["pose_rotation_change_rad"] for p in per)
        tr = max(p["pose_translation_change_m"] for p in per)
        out[name] = {"assumed": assumed, "alternative": alternative, "per_pair": per,
                     "max_pose_rotation_change_rad": rot, "max_pose_translation_change_m": tr,
                     "beyond_parity_tolerance": bool(rot > 1e-4 or tr > 1e-3)}
    return out


def main():
    res = {"shape": [W, H], "pairs": [list(p) for p in PAIRS],
           "note": "oracle with ONE assumption about OpenCV 4.5.4 replaced by its alternative vs the oracle as it stands; "
                   "parity tolerance: 1e-4 rad / 1e-3 m per relative pose",
           "assumptions": measure()}
    with open(OUT, "w") as f:
        json.dump(res, f, indent=1)
    for k, v in res["assumptions"].items():
        p = v["per_pair"][0]
        print("%-24s disp px %7d (%.4f of valid)  klt status %4d  track %.4f px  pose %.2e rad %.2e m %s" % (
            k, p["disparity_pixels_changed"], p["disparity_pixels_changed_frac_of_valid"], p["klt_status_changed"],
            p["klt_max_track_change_px"], v["max_pose_rotation_change_rad"], v["max_pose_translation_change_m"],
            "BEYOND" if v["beyond_parity_tolerance"] else "within"))


if __name__ == "__main__":
    main()
