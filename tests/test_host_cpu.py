"""CPU-only checks of the shipped library: it loads, exports every symbol the header declares, fails loudly
without a GPU, and its host-side solvers (the part of estimateMotion that stays on the CPU) agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from visualodometry_amd import abi, sequence
from oracle import binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1.0]])


def header_functions():
    text = open(os.path.join(ROOT, "include", "vo_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vo_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = abi.lib()
    names = header_functions()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(abi.EXPORTS) == names


def test_default_params_are_the_reference_literals():
    p = abi.default_params(1241, 376)
    assert (p.num_disparities, p.block_size, p.p1, p.p2) == (96, 11, 968, 3872)   # Source/VOClass.cpp:238-254
    assert (p.fast_threshold, p.fast_nms) == (20, 1)                               # Source/VOClassFeatures.cpp:27-28
    assert (p.lk_win, p.lk_max_level, p.lk_max_count) == (15, 3, 50)               # Source/VOClassFeatures.cpp:69-102
    assert p.lk_epsilon == float(np.float32(0.03)) and p.loop_threshold == 2 and p.depth_thresh == 3000
    assert (p.ransac_iters, p.ransac_reproj, p.ransac_conf) == (100, 8.0, 0.99)
    op = ob.default_params(1241, 376)
    for f, _ in ob.Params._fields_:
        assert getattr(op, f) == getattr(p, f), f


def test_no_gpu_means_loud_failure(gpu_available):
    if gpu_available:
        pytest.skip("a GPU is present")
    with pytest.raises(abi.VoError) as e:
        abi.Context(abi.default_params(320, 96, num_disparities=64))
    assert e.value.code == -4


def test_null_arguments_are_rejected():
    lib = abi.lib()
    assert lib.vo_default_params(None, 10, 10) == -1
    assert lib.vo_set_calib(None, None, None) == -1
    assert lib.vo_integrate_pose(None, None) == -1
    h = C.c_void_p()
    assert lib.vo_create(None, 0, C.byref(h)) == -1 and b"null" in lib.vo_create_error()


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def scene(n, seed, noise=0.0):
    rng = np.random.RandomState(seed)
    obj = np.stack([rng.uniform(-8, 8, n), rng.uniform(-2, 2, n), rng.uniform(5, 40, n)], axis=1).astype(np.float32)
    r = rng.normal(0, 0.02, 3); t = np.array([0.05, -0.02, -0.8]) + rng.normal(0, 0.05, 3)
    R, _ = ob.rodrigues(r)
    cam = obj.astype(np.float64) @ R.T + t
    img = np.stack([cam[:, 0] / cam[:, 2] * K[0, 0] + K[0, 2], cam[:, 1] / cam[:, 2] * K[1, 1] + K[1, 2]], axis=1)
    return obj, (img + rng.normal(0, noise, img.shape)).astype(np.float32)


def test_host_epnp_matches_oracle_bitwise():
    lib = abi.lib()
    Kc = np.ascontiguousarray(K)
    for seed in range(40):
        obj, img = scene(5, 100 + seed, noise=0.5)   # the minimal 5-point case RANSAC uses (2-dim null space)
        r, t = np.zeros(3), np.zeros(3)
        assert lib.vo_host_epnp(_ptr(obj), _ptr(img), 5, _ptr(Kc), _ptr(r), _ptr(t)) == 0
        orr, ot = ob.epnp(obj, img, K)
        assert np.array_equal(r, orr) and np.array_equal(t, ot), seed
    obj, img = scene(80, 7, noise=0.3)
    r, t = np.zeros(3), np.zeros(3)
    assert lib.vo_host_epnp(_ptr(obj), _ptr(img), 80, _ptr(Kc), _ptr(r), _ptr(t)) == 0
    orr, ot = ob.epnp(obj, img, K)
    assert np.array_equal(r, orr) and np.array_equal(t, ot)


def test_fixed_size_epnp_is_the_host_solver_bit_for_bit():
    """epnp5.h is what the GPU runs with pnp_device = 1; compiled for the host it must be pnp_host::epnp's arithmetic."""
    lib = abi.lib()
    Kc = np.ascontiguousarray(K)
    for seed in range(300):
        obj, img = scene(5, 500 + seed, noise=0.1 * (seed % 9))
        r, t, rf, tf = np.zeros(3), np.zeros(3), np.zeros(3), np.zeros(3)
        assert lib.vo_host_epnp(_ptr(obj), _ptr(img), 5, _ptr(Kc), _ptr(r), _ptr(t)) == 0
        assert lib.vo_host_epnp5_fixed(_ptr(obj), _ptr(img), _ptr(Kc), _ptr(rf), _ptr(tf)) == 0
        assert np.array_equal(r, rf) and np.array_equal(t, tf), seed
    assert lib.vo_host_epnp5_fixed(None, _ptr(img), _ptr(Kc), _ptr(rf), _ptr(tf)) == -1


def test_host_iterative_matches_oracle_bitwise():
    lib = abi.lib()
    Kc = np.ascontiguousarray(K)
    for n, seed in ((6, 1), (40, 2), (900, 3)):
        obj, img = scene(n, seed, noise=0.4)
        o64, i64 = obj.astype(np.float64), img.astype(np.float64)
        r, t = np.zeros(3), np.zeros(3)
        rc = lib.vo_host_pnp_iterative(_ptr(o64), _ptr(i64), n, _ptr(Kc), _ptr(r), _ptr(t))
        orc, orr, ot = ob.pnp_iterative(o64, i64, K)
        assert rc == orc == 1 and np.array_equal(r, orr) and np.array_equal(t, ot)
    o64, i64 = obj[:5].astype(np.float64), img[:5].astype(np.float64)
    assert lib.vo_host_pnp_iterative(_ptr(o64), _ptr(i64), 5, _ptr(Kc), _ptr(r), _ptr(t)) == -3


def test_host_rng_and_rodrigues_match_oracle():
    lib = abi.lib()
    for count in (6, 53, 4187):
        out = np.zeros((100, 5), dtype=np.int32)
        assert lib.vo_host_rng_subsets(count, 100, _ptr(out)) == 0
        assert np.array_equal(out, ob.rng_subsets(count, 100))
    rng = np.random.RandomState(0)
    for _ in range(10):
        r = rng.normal(0, 0.7, 3)
        R = np.zeros((3, 3))
        assert lib.vo_host_rodrigues(_ptr(r), _ptr(R)) == 0
        assert np.array_equal(R, ob.rodrigues(r)[0])


def test_integrate_pose_is_pose_times_inverse():
    rng = np.random.RandomState(4)
    pose = np.eye(4)
    ref = np.eye(4)
    for _ in range(50):
        R, _ = ob.rodrigues(rng.normal(0, 0.05, 3))
        Rt = np.eye(4); Rt[:3, :3] = R; Rt[:3, 3] = rng.normal(0, 0.5, 3)
        pose = abi.integrate_pose(pose, Rt)
        ref = ref @ ob.inv4x4(Rt)
        assert np.allclose(pose, ref, rtol=0, atol=1e-13)
    assert np.allclose(pose[:3, :3] @ pose[:3, :3].T, np.eye(3), atol=1e-12)


def test_sequence_helpers():
    lengths = [4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201]
    for world in (1, 2, 4, 8):
        own = sequence.assign_sequences(lengths, world)
        assert sorted(i for o in own for i in o) == list(range(11))
        loads = [sum(lengths[i] - 1 for i in o) for o in own]
        assert max(loads) <= max(max(lengths), sum(lengths) / world * 1.34)
    Rt = np.eye(4); Rt[2, 3] = -0.8
    traj, pose = sequence.integrate([Rt, Rt, Rt])
    assert np.allclose(traj[:, 2], [0, 0.8, 1.6, 2.4])
    assert sequence.trajectory_rmse(traj, traj) == 0.0
    est = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 1.0]]); gt = np.array([[0, 0, 0], [1, 0, 1], [2, 0, 1.0]])
    assert sequence.trajectory_rmse(est, gt) == ob.trajectory_rmse(est, gt)


def test_cpp_voclass_example_is_built_and_links():
    """examples/main_loop (the reference's main loop on include/VOClass.h) is built by __graft_entry__.build()."""
    import subprocess
    exe = os.path.join(ROOT, "examples", "main_loop")
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build()
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


def test_abi_header_is_plain_c():
    """include/vo_abi.h is the boundary: it must compile as C99 on its own (no C++ or torch types in any signature)"""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                        os.path.join(root, "include", "vo_abi.h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
