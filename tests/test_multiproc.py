"""The N > 1 path on CPU: two processes over gloo.  Sequences are sharded (LPT), every rank produces the relative
poses of its own sequences, and ONE all_gather of fixed-size padded buffers gives every rank the complete result;
integrating on the host then yields the same trajectories as a single process."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from visualodometry_amd import sequence


def fake_sequence_Rts(seq_id, steps):
    """Deterministic stand-in for the per-step [R|t] a GPU would produce (the collective does not care)."""
    rng = np.random.RandomState(100 + seq_id)
    out = np.zeros((steps, 16))
    for k in range(steps):
        a = rng.normal(0, 0.02)
        Rt = np.eye(4)
        Rt[0, 0] = Rt[2, 2] = np.cos(a); Rt[0, 2] = np.sin(a); Rt[2, 0] = -np.sin(a)
        Rt[:3, 3] = [rng.normal(0, 0.01), 0, -0.8]
        out[k] = Rt.reshape(16)
    return out


LENGTHS = [9, 4, 12, 3, 2, 7]


def worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    owner = sequence.assign_sequences(LENGTHS, world)
    local = {i: fake_sequence_Rts(i, LENGTHS[i] - 1) for i in owner[rank]}
    allr = sequence.gather_trajectories(local, len(LENGTHS), max(LENGTHS) - 1)
    finals = {i: sequence.integrate(allr[i].reshape(-1, 4, 4))[0][-1].tolist() for i in sorted(allr)}
    q.put((rank, owner, finals))
    dist.barrier()
    dist.destroy_process_group()


def chunk_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    owner = sequence.assign_chunks(LENGTHS, world)
    local = {}
    for i, first, end in owner[rank]:
        local[i] = (first, fake_sequence_Rts(i, LENGTHS[i] - 1)[first:end])   # the slice this rank would have computed
    allr = sequence.gather_trajectories(local, len(LENGTHS), max(LENGTHS) - 1)
    finals = {i: sequence.integrate(allr[i].reshape(-1, 4, 4))[0][-1].tolist() for i in sorted(allr)}
    q.put((rank, owner, finals))
    dist.barrier()
    dist.destroy_process_group()


def cloud_of(rank, i):
    rng = np.random.RandomState(10 * rank + i)
    return rng.normal(0, 5, size=(3 + 2 * i + rank, 3)).astype(np.float32)


def cloud_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    got = sequence.gather_point_clouds([cloud_of(rank, i) for i in range(3)], cap=16)
    ok = all(np.array_equal(got[r][i], cloud_of(r, i)) for r in range(world) for i in range(3))
    # the same sets handed over already padded into one tensor (what a rank does whose points sit in GPU memory); the ranks'
    # paddings differ, so one of them has to widen its buffer to the other's
    import torch
    mine = [cloud_of(rank, i) for i in range(3)]
    k = max(a.shape[0] for a in mine) + 2 * rank
    buf = torch.full((3, k, 3), -7.0)
    for i, a in enumerate(mine):
        buf[i, :a.shape[0]] = torch.from_numpy(a)
    got = sequence.gather_point_clouds((buf, [a.shape[0] for a in mine]), cap=16, reserve=16)
    ok = ok and all(np.array_equal(got[r][i], cloud_of(r, i)) for r in range(world) for i in range(3))
    q.put((rank, ok, len(got)))
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sequence_sharding_and_gather_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    owner = res[0][1]
    assert sorted(i for o in owner for i in o) == list(range(len(LENGTHS)))
    assert set(owner[0]).isdisjoint(owner[1])
    # every rank ends up with every sequence, identical to the single-process result
    want = {i: sequence.integrate(fake_sequence_Rts(i, LENGTHS[i] - 1).reshape(-1, 4, 4))[0][-1].tolist() for i in range(len(LENGTHS))}
    for rank, _, finals in res:
        assert finals == want, rank


def test_gather_single_process_is_identity():
    local = {0: fake_sequence_Rts(0, 5), 2: fake_sequence_Rts(2, 3)}
    out = sequence.gather_trajectories(local, 3, 8)
    assert sorted(out) == [0, 2] and np.array_equal(out[0], local[0]) and np.array_equal(out[2], local[2])


def test_chunk_assignment_is_a_balanced_partition():
    for world in (1, 2, 3, 8):
        owner = sequence.assign_chunks(LENGTHS, world)
        seen = {}
        for r, chunks in enumerate(owner):
            for i, first, end in chunks:
                assert 0 <= first < end <= LENGTHS[i] - 1
                for k in range(first, end):
                    assert (i, k) not in seen
                    seen[(i, k)] = r
        total = sum(n - 1 for n in LENGTHS)
        assert len(seen) == total
        loads = [sum(e - f for _, f, e in chunks) for chunks in owner]
        assert max(loads) - min(loads) <= 1
    # KITTI odometry lengths on 8 ranks: whole sequences are bounded by the longest one, chunks are not
    kitti = [4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201]
    whole = sequence.assign_sequences(kitti, 8)
    assert max(sum(kitti[i] - 1 for i in o) for o in whole) == 4660
    chunks = sequence.assign_chunks(kitti, 8)
    assert max(sum(e - f for _, f, e in c) for c in chunks) == -(-(sum(kitti) - len(kitti)) // 8)


def test_chunk_sharding_and_gather_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=chunk_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owner = sorted(res)[0][1]
    assert any(i == j for i, _, _ in owner[0] for j, _, _ in owner[1])   # one sequence is split between the ranks
    want = {i: sequence.integrate(fake_sequence_Rts(i, LENGTHS[i] - 1).reshape(-1, 4, 4))[0][-1].tolist() for i in range(len(LENGTHS))}
    for rank, _, finals in res:
        assert finals == want, rank


def test_point_cloud_gather_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=cloud_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True, 2), (1, True, 2)]
    single = sequence.gather_point_clouds([cloud_of(0, 0)], cap=8)
    assert len(single) == 1 and np.array_equal(single[0][0], cloud_of(0, 0))
    import pytest
    with pytest.raises(ValueError):
        sequence.gather_point_clouds([cloud_of(0, 2)], cap=4)
