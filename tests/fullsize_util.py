"""Helpers shared by the full-size fixture generators' tests: array hashing and parallel rendering of synthetic frames."""
import hashlib
import os

import numpy as np

from visualodometry_amd import synth


def h64(a):
    """64-bit BLAKE2b digest of an array's bytes (C order) as an unsigned integer (the hash of tests/golden/make_kitti200.py)."""
    return int.from_bytes(hashlib.blake2b(np.ascontiguousarray(a).tobytes(), digest_size=8).digest(), "little")


def _render(job):
    seed, w, h, i, kw = job
    return synth.Sequence(seed=seed, width=w, height=h, **kw).frame(i)


def render_frames(seed, w, h, count, **kw):
    """`count` consecutive stereo frames of a synthetic sequence, rendered on forked worker processes (numpy only: the
    children never touch a GPU, and they leave through os._exit)."""
    try:
        ncpu = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        ncpu = os.cpu_count() or 1
    workers = max(1, min(16, ncpu, count))
    jobs = [(seed, w, h, i, kw) for i in range(count)]
    if workers == 1:
        return [_render(j) for j in jobs]
    import multiprocessing as mp
    with mp.get_context("fork").Pool(workers) as pool:
        return pool.map(_render, jobs, chunksize=1)


def rotation_angle(Ra, Rb):
    """angle (rad) of the rotation that takes Ra to Rb"""
    f = np.linalg.norm(Ra.T @ Rb - np.eye(3))          # Frobenius norm of (R - I) = 2 sqrt(2) sin(angle / 2)
    return float(2.0 * np.arcsin(min(1.0, f / (2.0 * np.sqrt(2.0)))))
