"""Sequence driver and sequence-level sharding.

The reference's main loop (Source/main.cpp:30-65) walks frame pairs (t, t+1) of one sequence and integrates
poseRt = poseRt * Rt^-1 (Source/VOClass.cpp:587).  A pair reads nothing from the previous pair except poseRt,
so pairs are independent up to an ordered product of 4x4 matrices: run_sequence() pushes B consecutive pairs
through the GPU per call and does the ordered product on the host with the library's LU inverse, which gives the
same trajectory as the one-pair-at-a-time loop.

Across GPUs the unit of work is a whole sequence (or a contiguous chunk of one): assign_sequences() is
longest-processing-time-first; gather_trajectories() is the single collective of a run (RCCL on GPUs, gloo in
the CPU tests), after all frames are done.
"""
import numpy as np

from . import abi


def assign_sequences(lengths, world_size):
    """LPT assignment: returns a list (one entry per rank) of sequence indices; deterministic."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    load = [0] * world_size
    owner = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owner[r].append(i)
        load[r] += max(lengths[i] - 1, 0)  # a sequence of n frames has n-1 steps
    return owner


def assign_chunks(lengths, world_size):
    """Chunked sharding: the steps of all sequences, laid end to end, are cut into world_size contiguous ranges of
    (almost) equal size.  Returns per rank a list of (sequence index, first step, end step) with end exclusive; step k
    of a sequence is the frame pair (k, k + 1).  Unlike whole-sequence sharding the makespan is not bounded below by
    the longest sequence; the per-step [R|t] are position-independent, so the ordered product over the gathered
    steps is the same as a serial run's."""
    steps = [max(n - 1, 0) for n in lengths]
    total = sum(steps)
    owner = [[] for _ in range(world_size)]
    start = 0  # global index of the current sequence's step 0
    for i, n in enumerate(steps):
        for r in range(world_size):
            lo = max(r * total // world_size, start)
            hi = min((r + 1) * total // world_size, start + n)
            if hi > lo:
                owner[r].append((i, lo - start, hi - start))
        start += n
    return owner


def run_chunk(ctx, frames, first, end, batch=None):
    """Steps first..end-1 of a sequence (frame pairs (k, k+1)): frames[first..end] are read.  Same results as the
    corresponding slice of run_sequence()."""
    Rt, info = run_sequence(ctx, frames[first:end + 1], batch)
    return Rt, info


def integrate(Rts, pose0=None):
    """Ordered product pose_k = pose_{k-1} * inv(Rt_k); returns (k+1) x 3 translations starting at pose0's."""
    pose = np.eye(4) if pose0 is None else np.array(pose0, dtype=np.float64)
    traj = [pose[:3, 3].copy()]
    for Rt in Rts:
        pose = abi.integrate_pose(pose, Rt)
        traj.append(pose[:3, 3].copy())
    return np.array(traj), pose


def trajectory_rmse(estimated, ground_truth):
    """VOClass::computeErrorInPoseEstimation (Source/VOClass.cpp:627-649): float accumulator, RMSE of translations."""
    err = np.float32(0)
    for e, g in zip(np.asarray(estimated, dtype=np.float64), np.asarray(ground_truth, dtype=np.float64)):
        err = np.float32(err + ((g[0] - e[0]) ** 2 + (g[1] - e[1]) ** 2 + (g[2] - e[2]) ** 2))
    return float(np.float32(np.sqrt(np.float32(err / len(estimated)))))


def run_sequence(ctx, frames, batch=None):
    """frames: list of (left, right) u8 images.  Returns (relative Rt per step [n-1,4,4], frame infos)."""
    n = len(frames)
    batch = batch or ctx.params.max_batch
    Rts, infos = [], []
    for s in range(0, n - 1, batch):
        e = min(s + batch, n - 1)
        # frames s..e: pair i is (frame i, frame i + 1); each frame is uploaded and pyramided once
        ctx.set_sequence(np.stack([frames[i][0] for i in range(s, e + 1)]), np.stack([frames[i][1] for i in range(s, e + 1)]))
        Rt, info = ctx.step()
        Rts.extend(Rt)
        infos.extend(info)
    return np.array(Rts), infos


def gather_trajectories(local, n_sequences, max_steps, group=None):
    """local: {sequence index: [steps, 16] relative Rt}, or {sequence index: (first step, [steps, 16])} for a rank that
    holds a chunk of a sequence (several ranks may then hold disjoint chunks of one sequence).  One all_gather of
    fixed-size padded buffers; returns {sequence index: [steps, 16]} for all sequences on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    buf = torch.zeros((n_sequences, max_steps, 16), dtype=torch.float64)
    cnt = torch.zeros((n_sequences,), dtype=torch.int64)
    for i, Rt in local.items():
        first, Rt = Rt if isinstance(Rt, tuple) else (0, Rt)
        a = np.asarray(Rt, dtype=np.float64).reshape(-1, 16)
        buf[i, first:first + a.shape[0]] = torch.from_numpy(a)
        cnt[i] = a.shape[0]
    if dist.is_initialized():   # a one-rank group runs the same collective on the same device tensors
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        buf, cnt = buf.to(dev), cnt.to(dev)
        bufs = [torch.empty_like(buf) for _ in range(world)]
        cnts = [torch.empty_like(cnt) for _ in range(world)]
        dist.all_gather(bufs, buf, group=group)
        dist.all_gather(cnts, cnt, group=group)
        buf = torch.stack(bufs).sum(0).cpu()   # every step is owned by exactly one rank; the others hold zeros there
        cnt = torch.stack(cnts).sum(0).cpu()
    return {i: buf[i, :int(cnt[i])].numpy().copy() for i in range(n_sequences) if int(cnt[i]) > 0}


_pinned = {}


def _host_landing(shape, dtype, device):
    """a host tensor for results coming off `device`: page-locked (and kept for the next call) when they come from a GPU"""
    import torch
    if device.type != "cuda":
        return torch.empty(shape, dtype=dtype)
    need = 1
    for d in shape:
        need *= int(d)
    key = (dtype, device.index)
    buf = _pinned.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.empty(max(need, 1), dtype=dtype, pin_memory=True)
        _pinned[key] = buf
    return buf[:need].view(shape)


def gather_point_clouds(clouds, cap, group=None, reserve=0):
    """clouds: this rank's list of [k_i, 3] float32 point sets (the triangulated points of its frame pairs, k_i <= cap), the
    same number of sets on every rank; or a pair (buf, counts) with the sets already padded into ONE [n, k, 3] float32 torch
    tensor (on the collective's device: the points then never visit the host before the gather) and their counts.  One
    all_gather of the counts, then one of fixed-size buffers padded to the largest count of any rank (not to `cap`: a
    KITTI-shape frame has about 4 000 points for a capacity of 16 384); returns, on every rank, a list over ranks of lists of
    [k_i, 3] arrays.  reserve: points per set to size the page-locked landing buffer for (a first call made for set-up passes
    the capacity, so that later calls allocate nothing)."""
    import torch
    import torch.distributed as dist
    padded = isinstance(clouds, tuple)
    if padded:
        pbuf, counts = clouds
        n = len(counts)
        if pbuf.dim() != 3 or pbuf.shape[0] != n or pbuf.shape[2] != 3 or pbuf.dtype != torch.float32:
            raise ValueError("padded point sets must be one [n, k, 3] float32 tensor")
        counts = [int(c) for c in counts]
        if any(c > cap for c in counts):
            raise ValueError("a point set has %d points, capacity %d" % (max(counts), cap))
        if not dist.is_initialized():
            host = pbuf.cpu().numpy()
            return [[host[i, :min(counts[i], host.shape[1])].copy() for i in range(n)]]
    else:
        n = len(clouds)
        arrs = [np.asarray(pts, dtype=np.float32).reshape(-1, 3) for pts in clouds]
        for i, a in enumerate(arrs):
            if a.shape[0] > cap:
                raise ValueError("point set %d has %d points, capacity %d" % (i, a.shape[0], cap))
        if not dist.is_initialized():
            return [[a.copy() for a in arrs]]
        counts = [a.shape[0] for a in arrs]
    world = dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    cnt = torch.tensor(counts, dtype=torch.int64).to(dev)
    cnts = [torch.empty_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    cnts = [c.cpu() for c in cnts]
    kmax = max(1, max(int(c.max()) if c.numel() else 0 for c in cnts))
    if padded:
        buf = pbuf.to(dev)
        if buf.shape[1] > kmax:
            buf = buf[:, :kmax]
        elif buf.shape[1] < kmax:   # another rank has a larger set
            buf = torch.nn.functional.pad(buf, (0, 0, 0, kmax - buf.shape[1]))
        buf = buf.contiguous()
    else:
        buf = torch.zeros((n, kmax, 3), dtype=torch.float32)
        for i, a in enumerate(arrs):
            buf[i, :a.shape[0]] = torch.from_numpy(a)
        buf = buf.to(dev)
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    if reserve:
        _host_landing((world, n, max(reserve, kmax), 3), torch.float32, dev)
    host = _host_landing((world, n, kmax, 3), torch.float32, dev)
    for r in range(world):
        host[r].copy_(bufs[r], non_blocking=True)
    if dev.type == "cuda":
        torch.cuda.current_stream().synchronize()
    host = host.numpy().copy()   # the landing buffer is reused by the next call
    return [[host[r, i, :int(cnts[r][i])] for i in range(n)] for r in range(world)]
