#!/usr/bin/env python3
"""Runs one KITTI-shape frame pair through the library named by VO_AMD_LIB (default: the in-tree build) and writes the
stage outputs to an .npz, so that two builds of a kernel can be compared bit for bit on a GPU box:
    VO_AMD_LIB=$PWD/gpurun_variants/libvo_prev.so python3 tools/dump_stages.py /tmp/a.npz && python3 tools/dump_stages.py /tmp/b.npz
    python3 tools/dump_stages.py --compare /tmp/a.npz /tmp/b.npz
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def dump(path, W=1241, H=376):
    from visualodometry_amd import abi, synth
    seq = synth.Sequence(seed=1000, width=W, height=H)
    f0, f1 = seq.frame(3), seq.frame(4)
    p = abi.default_params(W, H, max_batch=1, max_features=8192)
    ctx = abi.Context(p)
    ctx.set_calib(seq.P0, seq.P1)
    ctx.set_images(f0[0], f0[1], f1[0], f1[1])
    ctx.compute_disparity()
    disp = ctx.disparity16(0)
    ctx.detect_fast()
    feats = ctx.features(0)
    ctx.match_klt()
    legs, st = ctx.klt_legs(len(feats))
    m1, m2 = ctx.matches(0)
    Rt, info = ctx.step()
    np.savez(path, disp=disp, feats=feats, legs=np.asarray(legs), status=np.asarray(st), m1=m1, m2=m2, Rt=Rt[0])
    ctx.close()


def compare(a, b):
    A, B = np.load(a), np.load(b)
    bad = 0
    for k in A.files:
        same = A[k].shape == B[k].shape and np.array_equal(A[k].view(np.uint8), B[k].view(np.uint8))
        print("%-8s %s %s" % (k, A[k].shape, "identical" if same else "DIFFERENT"))
        bad += not same
    return bad


if __name__ == "__main__":
    if sys.argv[1] == "--compare":
        sys.exit(compare(sys.argv[2], sys.argv[3]))
    dump(sys.argv[1])
