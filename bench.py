#!/usr/bin/env python3
"""Throughput benchmark of the stereo front-end hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--batch B] [--workload kitti|4k|kitti11|dense]

One "step" = one pass of the whole per-frame hot path (disparity, FAST, circular KLT, triangulation,
PnP-RANSAC; one iteration of the reference's main loop, Source/main.cpp:36-60) over a batch of B independent
synthetic stereo frame pairs that are already resident in HBM.

Workloads
  kitti, 4k   weak scaling: every rank streams its own rendered sequence (seed 1000 + rank) through K steps of B
              consecutive frame pairs; no pair repeats inside a batch or between the batches in flight.
  dense       the dense-tracking configuration at KITTI shape: 20 000 corners per frame (threshold 10, no suppression, the first
              20 000 in row-major order), 21x21 tracking window on a 4-level pyramid, 1 024 RANSAC hypotheses solved and scored
              per frame (one scoring launch per batch).
  kitti11     strong scaling, the 11 KITTI 00-10-shaped sequences (4541 ... 1201 frames): whole sequences are dealt
              to ranks longest-first (sequence.assign_sequences), or with --chunked the steps of all sequences are
              cut into equal contiguous ranges (sequence.assign_chunks); value = all steps / slowest rank's time.
Independent units, no collective inside the path; the relative poses (and the last batch's point clouds) are gathered
once at the end with RCCL.  Prints ONE JSON line on rank 0.

Ranks: under a launcher (torchrun) RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment.  Without one,
`--gpus N` with N > 1 makes this process start N children itself (one per GPU, before anything here touches a GPU) and
relay rank 0's line.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
KITTI_SEQ_LENGTHS = (4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201)
METRIC = "stereo frames/sec at 1241x376 (1/2/4/8 GPU) + achieved HBM GB/s vs roofline"
# what bounds each kernel that has a byte model: the HBM roofline is read against the longest
# kernel of the "hbm" class; the tracker is bound by vector-instruction issue and is reported beside it
KERNEL_BOUND = {"sgbm_prefilter": "hbm", "sgbm_cost_vpath": "hbm", "sgbm_hpath_ckpt": "hbm", "sgbm_hpath_resolve": "hbm",
                "sgbm_lrcheck_median": "hbm", "lk": "valu"}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def workload_shape(name, args):
    if name in ("kitti", "kitti11"):
        s = dict(width=1241, height=376, num_disparities=96, lk_win=15, max_features=16384)
    elif name == "4k":
        # the pixel-matched 4K scene has the corner density per pixel of the KITTI-shape one: about 77 000 corners a frame
        s = dict(width=3840, height=2160, num_disparities=256, lk_win=15, max_features=98304)
    elif name == "dense":
        s = dict(width=1241, height=376, num_disparities=96, lk_win=21, max_features=20000,
                 extra=dict(fast_threshold=10, fast_nms=0, ransac_iters=1024, ransac_chunk=1024,
                            # 43 008 minimal solves per step outrun 16 host threads (77 ms against 30 ms of GPU work): they run on the GPU
                            pnp_device=1 if args.pnp_device is None else args.pnp_device))
    else:
        raise ValueError(name)
    if args.width:
        s["width"] = args.width
    if args.height:
        s["height"] = args.height
    if args.disparities:
        s["num_disparities"] = args.disparities
    return s


def stripe_rows(H, block=11):
    """Rows of the per-stripe cost volume: 4 fixed stripes, each with its warm-up overlap rows."""
    sz = -(-H // 4)
    overlap = (block // 2 + 1) + int(np.ceil(0.1 * sz))
    return sum(max(min((s + 1) * sz, H) - max(min(s * sz - overlap, H), 0), 0) for s in range(4))


def pyramid_pixels(W, H, levels=4):
    pyr, w, h = 0, W, H
    for _ in range(levels):
        pyr += w * h
        w, h = (w + 1) // 2, (h + 1) // 2
    return pyr


def kernel_bytes(W, H, D, srows, nframes):
    """Compulsory (algorithmic) bytes each kernel moves per launch: every input and output counted once."""
    w1 = W - D
    P = W * H
    pyr = pyramid_pixels(W, H)
    vol_h = H * w1 * D * 2          # the rows of a volume that produce output (no stripe warm-up rows)
    seg = 16 if D // 32 <= 4 else 8
    ckpt = H * (-(-w1 // seg)) * 16 * (D // 32 + 1) * 4
    per = {
        "sgbm_prefilter": 2 * P + 2 * P * 8,
        "sgbm_cost_vpath": 2 * P * 8 + 2 * vol_h,       # prefiltered rows in; C and L_top out
        "sgbm_hpath_ckpt": vol_h + ckpt,                # C in; path state at segment starts out
        "sgbm_hpath_resolve": 2 * vol_h + ckpt + H * W * 4,   # C, L_top, checkpoints in; packed winner out
        "sgbm_lrcheck_median": H * W * 4 + P * 2 + P * 2 + P * 4,   # packed winners in; pre-median, s16 and f32 maps out
        "lk": 4 * (pyr + 4 * pyr + pyr),                # 4 legs x (previous image + its derivatives + next image)
    }
    return {k: v * nframes for k, v in per.items()}


def frame_bytes(W, H, D, n_feat=4000):
    """Algorithmic bytes per stereo frame of the whole path: A = A_io + A_cv (one write + one read of the
    aggregated s16 cost volume), with P = W*H, Pyr = pixels of pyramid levels 0..3, C = H*(W-D)*D."""
    P = W * H
    pyr = pyramid_pixels(W, H)
    C = H * (W - D) * D
    a_io = 2 * P + (2 * P + 2 * P) + (P + 8 * n_feat) + 2 * pyr + 2 * (pyr + 4 * pyr) + 4 * (pyr + 4 * pyr + pyr)
    a_cv = 2 * 2 * C
    return a_io + a_cv


# ------------------------------------------------------------------ rendering (host, before any GPU use)
def scene_kwargs(workload, W):
    """The synthetic scene of a workload: the 4K one keeps the per-pixel statistics of the KITTI-shape scene (texture
    wavelengths, baseline, step and yaw scaled by 1241 / width), so that its tracking and pose load is representative:
    85 % of the corners close the matching circle, as at KITTI shape, where the unscaled scene kept a third."""
    from visualodometry_amd import synth
    return synth.pixel_matched_scene(W) if workload == "4k" else {}


def _render_one(job):
    from visualodometry_amd import synth
    seed, W, H, i, kw = job
    return synth.Sequence(seed=seed, width=W, height=H, **kw).frame(i)


def render_frames(seed, W, H, count, workers, cache_dir="", scene=None):
    """`count` consecutive stereo frames of the synthetic sequence `seed`, rendered on `workers` host processes.
    cache_dir: keep / reuse the rendered frames there (profiler passes, where the process must not fork)."""
    cache = os.path.join(cache_dir, "frames_s%d_%dx%d_n%d%s.npy" % (seed, W, H, count, "_pm" if scene else "")) if cache_dir else ""
    if cache and os.path.exists(cache):
        a = np.load(cache)
        return [(a[i, 0], a[i, 1]) for i in range(count)]
    jobs = [(seed, W, H, i, scene or {}) for i in range(count)]
    if workers <= 1 or count <= 2:
        out = [_render_one(j) for j in jobs]
    else:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, count)) as pool:   # forked before this process touches a GPU
            out = pool.map(_render_one, jobs, chunksize=1)
    if cache:
        os.makedirs(cache_dir, exist_ok=True)
        np.save(cache, np.stack([np.stack(f) for f in out]))
    return out


class stdout_to_stderr:
    """File descriptor 1 points to stderr inside the block: a communication library that prints a version banner from C
    code when its communicator comes up must not put it in front of the one JSON line that stdout carries."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


# ------------------------------------------------------------------ self-launch of N ranks
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args, argv):
    """Parent of an N-rank run without an outside launcher: starts N children (fresh processes, one per GPU) and relays
    rank 0's JSON line.  This process never initialises a GPU."""
    n = args.gpus
    env0 = dict(os.environ)
    env0.setdefault("MASTER_ADDR", "127.0.0.1")
    env0["MASTER_PORT"] = str(free_port())
    env0["WORLD_SIZE"] = str(n)
    env0["LOCAL_WORLD_SIZE"] = str(n)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        env = dict(env0)
        env["RANK"] = env["LOCAL_RANK"] = str(r)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0].decode()
    rcs = [p.wait() for p in procs]
    for line in out0.splitlines():   # the result line goes to stdout, anything a library printed there to stderr
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr)
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        log("[bench] ranks failed: %s" % bad)
        return 1
    return 0


# ------------------------------------------------------------------ plumbing stand-in for the multi-rank tests
class StubContext:
    """NOT a compute path: stands in for abi.Context in the multi-process plumbing test of this file (--stub-context), on
    machines without a GPU.  'Poses' are a deterministic function of the frames a step was handed (their byte sums), so the
    test can tell that every rank processed the frames it was assigned and that the gather put them in order."""

    def __init__(self, W, H, B):
        self.W, self.H, self.B, self.n = W, H, B, 0
        self._ptrs = None

    def set_calib(self, P0, P1):
        pass

    def set_images_device(self, n, ptrs, stride, frame_stride, keep=None):
        self.n, self._ptrs, self._fs, self._keep = n, [ptrs[0], ptrs[2]], frame_stride, keep

    def set_sequence_device(self, n, left, right, stride, frame_stride, keep=None):
        self.n, self._ptrs, self._fs, self._keep = n, [left, left + frame_stride], frame_stride, keep

    @staticmethod
    def pose_of(a_sum, b_sum):
        Rt = np.eye(4)
        Rt[:3, 3] = [(a_sum % 100003) * 1e-5, (b_sum % 100003) * 1e-5, -0.8]
        return Rt

    def _frame_sum(self, ptr):
        buf = (ctypes.c_uint8 * (self.W * self.H)).from_address(ptr)
        return int(np.frombuffer(buf, dtype=np.uint8).astype(np.int64).sum())

    def step_submit(self):
        pass

    def step_finish(self, strict=True):
        Rt = np.stack([self.pose_of(self._frame_sum(self._ptrs[0] + b * self._fs), self._frame_sum(self._ptrs[1] + b * self._fs))
                       for b in range(self.n)])
        info = [argparse.Namespace(n_features=0, n_matched=0, ok=1) for _ in range(self.n)]
        return Rt, info

    def step(self, strict=True):
        return self.step_finish()

    def object_points(self, b=0):
        return np.zeros((1, 3), dtype=np.float32), np.zeros((1, 2), dtype=np.float32)

    def synchronize(self):
        pass

    def set_timing(self, enable=True):
        pass

    def timing(self, reset=False):
        return {}


def cpu_baseline(frames, seq, shape, sample_pairs):
    """The CPU oracle (a port of the same path) timed on the host cores on a bounded sample of the workload."""
    from oracle import binding as ob
    threads = max(1, min(16, int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))))
    ob.set_threads(threads)
    extra = {k: v for k, v in shape.get("extra", {}).items() if k not in ("ransac_chunk", "pnp_device")}   # the port has no hypothesis batching
    p = ob.default_params(shape["width"], shape["height"], num_disparities=shape["num_disparities"], lk_win=shape["lk_win"], **extra)
    cap = shape["max_features"] if "extra" in shape else None
    pose = np.eye(4)
    nd = len(frames) - 1

    def one(i):
        a, b = frames[i], frames[i + 1]
        if cap is None:
            ob.step(a[0], a[1], b[0], b[1], p, seq.P0, seq.P1, pose)
            return
        # a capped corner list (the dense-tracking configuration keeps the first `cap` corners): the same stages one by one
        d32 = ob.disp_to_float(ob.sgbm(a[0], a[1], p))
        depth = ob.depth_map(d32, seq.P0[0, 0], seq.P1[0, 3] / (-seq.P0[0, 0]))
        kp = ob.fast(a[0], p.fast_threshold, bool(p.fast_nms), cap=cap)
        t1, t2 = ob.match_klt(a[0], a[1], b[0], b[1], kp, p)
        obj, _, i2 = ob.triangulate(t1, t2, depth, seq.P0)
        ob.solve_pnp_ransac(obj, i2, seq.P0, p)

    npairs = sample_pairs
    t0 = time.perf_counter()
    for k in range(npairs):
        one(k % nd)
    dt = time.perf_counter() - t0
    # the same port on one thread, on a smaller sample
    ob.set_threads(1)
    n1 = max(1, min(3, npairs))
    t1 = time.perf_counter()
    for k in range(n1):
        one(k % nd)
    dt1 = time.perf_counter() - t1
    ob.set_threads(threads)
    return {"value": npairs / dt, "unit": "stereo frames/s", "cores": threads, "kind": "port",
            "sample": "%d frame pairs of the same synthetic %dx%d sequence, %.1f s wall (SGBM on 4 stripe threads, LK on %d)" %
                      (npairs, shape["width"], shape["height"], dt, threads),
            "single_thread_value": n1 / dt1, "single_thread_sample": "%d frame pairs, %.1f s wall" % (n1, dt1)}


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300,
                    help="timed steps (default 300, about 2.3 s at KITTI shape: the fill and drain of the pipeline are 3 %% of 50 steps)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0, help="frame pairs per step (default: 56 at KITTI shape, 42 for dense tracking, 8 at 4K)")
    ap.add_argument("--workload", default="kitti", choices=["kitti", "4k", "kitti11", "dense"])
    ap.add_argument("--frames", type=int, default=0,
                    help="consecutive frames rendered per rank and kept in HBM (default: 200 at KITTI shape, the whole sequence of "
                         "the single-GPU configuration; otherwise one window of `batch` pairs per batch in flight, plus one)")
    ap.add_argument("--chunked", action="store_true", help="kitti11: contiguous step ranges per rank instead of whole sequences")
    ap.add_argument("--scale", type=float, default=1.0, help="kitti11: scale every sequence length (rehearsals, tests)")
    ap.add_argument("--cpu-pairs", type=int, default=0,
                    help="frame pairs of the CPU baseline's sample (default: 48 at KITTI shape, about 10 s on 16 threads; 1 at 4K)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--hw-queues", type=int, default=0, help="GPU_MAX_HW_QUEUES for this process (0: 8 for KITTI-shape runs of 100 steps and more and for kitti11, else 4; at most 32)")
    ap.add_argument("--pnp-device", type=int, default=None, choices=(0, 1),
                    help="RANSAC minimal solves on the host (0) or on the GPU (1); default: GPU for the dense workload, host otherwise")
    ap.add_argument("--inflight", type=int, default=0,
                    help="contexts taking turns (overlaps host PnP with GPU work; default 3, 4 for kitti11; at 4K a context holds "
                         "5.5 GB per frame pair)")
    ap.add_argument("--upload", action="store_true",
                    help="re-upload the batch's images from host memory before every step of the timed region (PCIe-inclusive rate; "
                         "the headline value is measured without it, inputs resident in HBM)")
    ap.add_argument("--pageable-upload", action="store_true", help="the upload-inclusive pass copies from pageable instead of page-locked host memory")
    ap.add_argument("--no-upload-leg", action="store_true", help="skip the second, upload-inclusive timed pass (config.value_with_upload)")
    ap.add_argument("--no-latency", action="store_true",
                    help="skip the pass that times one frame pair alone (batch of 1, 12 steps) after the run: use it for a kernel "
                         "profile that should hold launches of one batch size only")
    ap.add_argument("--latency", action="store_true", help="(kept for old command lines: the latency pass is on by default)")
    ap.add_argument("--no-collective", action="store_true",
                    help="single rank: do not create the one-rank RCCL communicator and skip the result gathers (by default they run "
                         "on device tensors exactly as with N ranks, after the timed steps)")
    ap.add_argument("--mode", default="sequence", choices=["pairs", "sequence"],
                    help="sequence (default): a step's window is B + 1 consecutive frames of the rank's sequence, handed over as such; "
                         "every frame is uploaded and pyramided once and serves as T2 of one pair and T1 of the next (same results). "
                         "pairs: the same window handed over as B independent frame pairs, 4 images and 4 pyramids each "
                         "(what the reference rebuilds on every step)")
    ap.add_argument("--rehearse", action="store_true",
                    help="more ranks than GPUs: rank r uses GPU r mod (GPUs present) and the gather runs over gloo (plumbing rehearsal)")
    ap.add_argument("--lenient", action="store_true", help="frames without a pose do not stop the run (timing-only kernel ablations, whose results are wrong on purpose)")
    ap.add_argument("--stub-context", action="store_true", help="plumbing test without a GPU: no kernels run, the value is meaningless")
    ap.add_argument("--dump-poses", default="", help="rank 0 writes the gathered relative poses to this .npz file")
    ap.add_argument("--frame-cache", default="", help="directory that keeps the rendered frames between runs (a run that finds "
                    "them there renders nothing and starts no worker processes: use it under a profiler)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--disparities", type=int, default=0)
    return ap.parse_args(argv)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args, argv)
    if args.batch <= 0:
        # Frame pairs per step.  The horizontal-path kernels run one wavefront per four image rows, 94 per KITTI-shape frame and two
        # per SIMD, so 42 frames fill the chip's 2048 wavefront slots almost twice over (96 %), where 16 frames leave a quarter
        # idle: that set the batch while a batch ran on one stream (5 250 frames/s at 42 against 4 790 at 16).  With the tracker
        # beside the horizontal passes the fit matters less and larger batches amortise more of a step's fixed costs: 56 pairs
        # give 5 985-5 995 frames/s against 5 890-5 925 at 42 over 300 steps, 5 795-5 825 against 5 540-5 575 over 20, and 5 826
        # against 5 659 on the 11-sequence workload (32 / 48 / 63 / 64 pairs: 5 965 / 5 958 / 6 074 / 6 068 over 200 steps).
        # Dense tracking stays at 42 (1 622 against 1 606 frames/s at 56), 4K x 256 at 8 (5.5 GB per pair in the batch).
        args.batch = 8 if args.workload == "4k" else 42 if args.workload == "dense" else 56
    if args.inflight <= 0:
        # Three batches in flight where the GPU is the limit (a fourth only adds contention).  The 11-sequence workload has the
        # longer host part per batch (4.2 ms against 2.8 ms: its frame pools are walked in both directions and the harder pairs
        # take more solver work), and a fourth batch covers it.
        args.inflight = 4 if args.workload == "kitti11" else 3
    # A process gets 4 hardware queues by default.  At the reference's parameters every batch in flight uses two streams
    # (disparity chain, tracking chain) and all share one high-priority stream for the pose solver's round trips; with more
    # streams than queues they serialise behind each other, so those runs ask for 8 queues (5 880 -> 5 995 frames/s; 2 per
    # batch beyond three batches; the box refuses more than 32).  The 4K and dense-tracking workloads keep a batch on one
    # stream, gain nothing from more queues (167.5 against 166, 1 553 against 1 546 frames/s) and keep 4.
    # (Under a profiler that initialises the runtime before main() the variable has to be exported in the shell instead.)
    # A short run is mostly fill and drain, where the extra queues cost instead (20 steps: 5 483 / 5 488 frames/s with 4 queues
    # against 5 409 / 5 368 with 8, and a lone pair takes 1.57 instead of 1.72 ms), so only runs of 100 steps or more ask for them.
    # The variable is set here whatever the environment holds (machines export HIP's default of 4); --hw-queues N chooses another.
    long_run = args.workload == "kitti11" or (args.workload == "kitti" and args.steps >= 100)
    queues = args.hw_queues if args.hw_queues > 0 else (max(8, 2 * args.inflight) if long_run else max(4, 2 * args.inflight if args.inflight > 3 else 4))
    os.environ["GPU_MAX_HW_QUEUES"] = str(min(32, queues))
    log("[bench] GPU_MAX_HW_QUEUES=%s for %d batches in flight" % (os.environ["GPU_MAX_HW_QUEUES"], args.inflight))

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    # host threads: one pool per process, sized to this rank's share of the node's cores
    try:
        ncpu = len(os.sched_getaffinity(0))   # the cores this process may run on, where the platform tells
    except (AttributeError, OSError):
        ncpu = os.cpu_count() or 1
    cpu_share = max(1, min(16, ncpu // max(local_world, 1)))
    if args.host_threads <= 0:
        args.host_threads = cpu_share

    shape = workload_shape(args.workload, args)
    W, H, D = shape["width"], shape["height"], shape["num_disparities"]
    B = args.batch
    nctx = max(1, args.inflight)
    strong = args.workload == "kitti11"

    # ---- input: rendered on the host before the GPU is touched (the worker pool is forked)
    from visualodometry_amd import synth
    seq_seed = 1000 if strong else 1000 + rank
    scene = scene_kwargs(args.workload, W)
    seq = synth.Sequence(seed=seq_seed, width=W, height=H, **scene)
    nframes = args.frames if args.frames > 0 else (200 if args.workload in ("kitti", "dense") else nctx * B + 1)
    nframes = max(nframes, B + 1)
    t0 = time.perf_counter()
    frames = render_frames(seq_seed, W, H, nframes, cpu_share, args.frame_cache, scene)
    log("[rank %d] rendered %d synthetic %dx%d stereo frames in %.1f s on %d host processes" % (rank, nframes, W, H, time.perf_counter() - t0, cpu_share))

    import torch
    import torch.distributed as dist
    stub = args.stub_context
    if not stub:
        from visualodometry_amd import abi
        ndev = torch.cuda.device_count()
        if ndev < 1:
            raise SystemExit("bench.py needs a GPU")
        if local_world > ndev and not args.rehearse:
            raise SystemExit("bench.py: %d ranks on this node but %d GPUs (use --rehearse to share GPUs in a plumbing rehearsal)" % (local_world, ndev))
        device_index = local_rank % ndev
    backend = "gloo" if (stub or args.rehearse) else "nccl"
    # A single rank still creates its (one-rank) RCCL communicator and runs the result gathers on device tensors, the same
    # code path as N ranks: the collective of this workload is then exercised by every run, not only on a multi-GPU node.
    collective = world > 1 or not (stub or args.no_collective)
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
        # RCCL ("nccl") is the backend of a real run; gloo only for rehearsing the multi-rank flow
        with stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
    if stub:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(device_index)
        dev = torch.device("cuda", device_index)

    def dev_sync():
        if not stub:
            torch.cuda.synchronize()

    # the whole rendered sequence lives in HBM: [nframes][H][W] per camera
    left = torch.from_numpy(np.stack([f[0] for f in frames])).to(dev)
    right = torch.from_numpy(np.stack([f[1] for f in frames])).to(dev)
    FS = W * H

    if stub:
        ctxs = [StubContext(W, H, B) for _ in range(nctx)]
    else:
        p = abi.default_params(W, H, max_batch=B, num_disparities=D, lk_win=shape["lk_win"], max_features=shape["max_features"],
                               host_threads=args.host_threads, **shape.get("extra", {}))
        if args.pnp_device is not None:
            p.pnp_device = args.pnp_device
        ctxs = [abi.Context(p, device_index) for _ in range(nctx)]
    for c in ctxs:
        c.set_calib(seq.P0, seq.P1)
    ctx = ctxs[0]

    def set_window(c, lptr, rptr, n):
        """n consecutive frame pairs starting at the frames lptr / rptr point to (device memory, frames FS bytes apart)"""
        if args.mode == "pairs":
            c.set_images_device(n, [lptr, rptr, lptr + FS, rptr + FS], W, FS, keep=(left, right))
        else:
            c.set_sequence_device(n, lptr, rptr, W, FS, keep=(left, right))

    # ---- weak-scaling workload: K steps over windows of the rank's own sequence
    # windows of B consecutive pairs that together cover every pair of the rendered sequence: disjoint ones from the start,
    # plus one that ends at the last frame (it overlaps its predecessor when B does not divide the pair count).  Batches in
    # flight at the same time work on different windows.
    win_start = list(range(0, nframes - B, B))
    if (nframes - 1) % B:
        win_start.append(nframes - 1 - B)
    nwin = len(win_start)
    host_left = host_right = None

    def submit_window(c, w, upload):
        f0 = win_start[w % nwin]
        if upload:   # host -> device copy of the window's frames, then the step
            if args.mode == "pairs":
                c.set_images(host_left[f0:f0 + B], host_right[f0:f0 + B], host_left[f0 + 1:f0 + B + 1], host_right[f0 + 1:f0 + B + 1])
            else:
                c.set_sequence(host_left[f0:f0 + B + 1], host_right[f0:f0 + B + 1])
        else:
            set_window(c, left.data_ptr() + f0 * FS, right.data_ptr() + f0 * FS, B)
        c.step_submit()

    def run_steps(k, upload=False):
        """k whole batches through the path, pipeline fill and drain included; returns the last batch's results"""
        out = None
        trace = [time.perf_counter()] if os.environ.get("VO_BENCH_TRACE") else None
        for s in range(min(nctx, k)):
            submit_window(ctxs[s], s, upload)
        for s in range(k):
            c = ctxs[s % nctx]
            out = c.step_finish(strict=not args.lenient)
            if trace is not None:
                trace.append(time.perf_counter())
            if s + nctx < k:
                submit_window(c, s + nctx, upload)
        if trace is not None:   # when each step's results were in hand, ms after the first submit
            log("[trace] steps finished at ms: " + " ".join("%.2f" % ((t - trace[0]) * 1e3) for t in trace[1:]))
        return out

    # ---- strong-scaling workload: the 11 KITTI-shaped sequences
    lengths = [max(2, int(round(n * args.scale))) for n in KITTI_SEQ_LENGTHS]
    from visualodometry_amd import sequence as vseq
    if strong:
        if args.chunked:
            mine = vseq.assign_chunks(lengths, world)[rank]
            loads = [sum(e - f for _, f, e in ch) for ch in vseq.assign_chunks(lengths, world)]
        else:
            owner = vseq.assign_sequences(lengths, world)
            mine = [(i, 0, lengths[i] - 1) for i in owner[rank]]
            loads = [sum(lengths[i] - 1 for i in o) for o in owner]
        period = 2 * (nframes - 1)

        def pool_index(i, t):
            """frame t of sequence i as an index into the rendered frames: a walk forth and back over them (every pair of
            consecutive frames is a real camera motion), started at a sequence-dependent phase"""
            m = (t + 7 * i) % period
            return m if m <= nframes - 1 else period - m
        staging = [(torch.empty((B + 1, H, W), dtype=torch.uint8, device=dev), torch.empty((B + 1, H, W), dtype=torch.uint8, device=dev))
                   for _ in range(nctx)]
        batches = []   # (sequence, first step, n pairs)
        for i, first, end in mine:
            for s in range(first, end, B):
                batches.append((i, s, min(B, end - s)))

        def submit_batch(k, j):
            i, s, n = batches[j]
            pi = [pool_index(i, s + q) for q in range(n + 1)]
            c = ctxs[k]
            if pi[-1] - pi[0] == n:   # an ascending run of the rendered frames: used where it lies
                lp, rp, keep = left.data_ptr() + pi[0] * FS, right.data_ptr() + pi[0] * FS, (left, right)
            else:                      # the walk turns inside the batch: its n + 1 frames are copied together (device to device)
                sl, sr = staging[k]
                idx = torch.tensor(pi, device=dev)
                torch.index_select(left, 0, idx, out=sl[:n + 1])
                torch.index_select(right, 0, idx, out=sr[:n + 1])
                if not stub:
                    torch.cuda.current_stream().synchronize()   # the context's stream does not wait for torch's
                lp, rp, keep = sl.data_ptr(), sr.data_ptr(), (sl, sr)
            if args.mode == "pairs":
                c.set_images_device(n, [lp, rp, lp + FS, rp + FS], W, FS, keep=keep)
            else:
                c.set_sequence_device(n, lp, rp, W, FS, keep=keep)
            c.step_submit()

        def run_batches(nb):
            local = {}
            out = None
            for j in range(min(nctx, nb)):
                submit_batch(j % nctx, j)
            for j in range(nb):
                out = ctxs[j % nctx].step_finish(strict=False)
                i, s, n = batches[j]
                local.setdefault(i, []).append((s, np.asarray(out[0]).reshape(n, 16)))
                if j + nctx < nb:
                    submit_batch(j % nctx, j + nctx)
            return local, out

    def barrier():
        if world > 1:
            dist.barrier()

    def gather_results(Rt, lastc, info):
        """the only collective of the weak-scaling workloads: the per-frame relative poses of every rank, and the point clouds
        (triangulated points) of the last batch as fixed-size padded buffers (result gather, RCCL over xGMI)"""
        tg = [time.perf_counter()]
        poses = torch.from_numpy(np.ascontiguousarray(Rt)).to(cdev)
        gathered = [torch.empty_like(poses) for _ in range(world)]
        dist.all_gather(gathered, poses)
        out = {r: g.cpu().numpy() for r, g in enumerate(gathered)}
        tg.append(time.perf_counter())
        if stub:
            mine = [lastc.object_points(b)[0] for b in range(B)]
        else:   # one strided device-to-device copy for the whole batch into the tensor the gather sends
            counts = lastc.object_points_batch_counts()
            kloc = max(1, max(counts))
            pts = torch.empty((B, kloc, 3), dtype=torch.float32, device=dev)
            lastc.object_points_batch_into(pts.data_ptr(), kloc)
            mine = (pts if cdev.type == "cuda" else pts.cpu(), counts)
        tg.append(time.perf_counter())
        clouds = vseq.gather_point_clouds(mine, cap=shape["max_features"])
        tg.append(time.perf_counter())
        if os.environ.get("VO_BENCH_GATHER_TIMES"):
            print("gather ms: poses %.2f, points off the GPU %.2f, clouds %.2f" % tuple(1e3 * (b - a) for a, b in zip(tg, tg[1:])), file=sys.stderr)
        assert len(clouds) == world
        dev_sync()
        return out, clouds

    # ---- warm-up
    if args.warmup > 0:
        if strong:
            run_batches(min(args.warmup, len(batches)))
        else:
            run_steps(args.warmup)
    for c in ctxs:
        c.set_timing(True)
        c.timing(reset=True)

    # ---- timed region
    cdev = dev if (world == 1 or backend == "nccl") else torch.device("cpu")
    if collective:
        # the collectives of the timed region once beforehand, with the same shapes and empty payloads: a communicator sets up
        # its channels and loads its kernels on first use, which is start-up cost, not part of a step
        with stdout_to_stderr():
            if strong:
                vseq.gather_trajectories({}, len(lengths), max(lengths) - 1)
            else:
                z = torch.zeros((B, 4, 4), dtype=torch.float64, device=cdev)
                dist.all_gather([torch.empty_like(z) for _ in range(world)], z)
                vseq.gather_point_clouds([np.zeros((0, 3), dtype=np.float32)] * B, cap=shape["max_features"], reserve=shape["max_features"])
            z1 = torch.zeros(1, dtype=torch.float64, device=cdev)
            dist.all_reduce(z1, op=dist.ReduceOp.MAX)
            dist.all_gather([torch.empty_like(z1) for _ in range(world)], z1)
            dev_sync()
    if args.upload:
        host_left, host_right = left.cpu().numpy(), right.cpu().numpy()
    barrier(); dev_sync()
    for c in ctxs:
        c.synchronize()
    t0 = time.perf_counter()
    gathered_poses = None
    if strong:
        local, last = run_batches(len(batches))
        Rt, info = last if last is not None else (np.zeros((0, 4, 4)), [])
        for c in ctxs:
            c.synchronize()
        dev_sync()
        t_compute = time.perf_counter() - t0
        loc = {}
        for i, parts in local.items():
            parts.sort(key=lambda x: x[0])
            loc[i] = (parts[0][0], np.concatenate([a for _, a in parts]))
        # the only collective: every rank ends up with every sequence's relative poses
        gathered_poses = vseq.gather_trajectories(loc, len(lengths), max(lengths) - 1)
        my_steps = sum(n for _, _, n in batches)
        total_pairs = sum(n - 1 for n in lengths)
    else:
        Rt, info = run_steps(args.steps, upload=args.upload)
        for c in ctxs:
            c.synchronize()
        dev_sync()
        t_compute = time.perf_counter() - t0
        clouds = None
        if collective:
            gathered_poses, clouds = gather_results(Rt, ctxs[(args.steps - 1) % nctx], info)
        else:
            gathered_poses = {0: np.asarray(Rt)}
        my_steps = args.steps * B
        total_pairs = world * B * args.steps
    barrier()
    dt = time.perf_counter() - t0
    t_all = [t_compute]
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tc = torch.tensor([t_compute], dtype=torch.float64, device=cdev)
        tcs = [torch.empty_like(tc) for _ in range(world)]
        dist.all_gather(tcs, tc)
        t_all = [float(t.item()) for t in tcs]

    if collective and world == 1 and not strong and not stub:   # one rank: the gather must hand back exactly what went in
        assert np.array_equal(gathered_poses[0], np.asarray(Rt)), "one-rank pose gather changed the poses"
        want = [ctxs[(args.steps - 1) % nctx].object_points(b)[0] for b in range(B)]
        assert all(np.array_equal(a, b_) for a, b_ in zip(clouds[0], want)), "one-rank point-cloud gather changed the points"

    timing = {}
    for c in ctxs:
        for k, v in c.timing().items():
            a = timing.get(k, (0.0, 0))
            timing[k] = (a[0] + v[0], a[1] + v[1])

    # ---- outside the timed region
    # the same K steps with every window re-uploaded from host memory first: the PCIe-inclusive rate
    # (page-locked host buffers, as a capture pipeline would hand them over; --pageable-upload for ordinary memory)
    value_with_upload = None
    if not strong and not args.upload and not args.no_upload_leg and not stub:
        if args.pageable_upload:
            host_left, host_right = left.cpu().numpy(), right.cpu().numpy()
        else:
            pinned = (left.cpu().pin_memory(), right.cpu().pin_memory())   # kept alive until the end of main()
            host_left, host_right = pinned[0].numpy(), pinned[1].numpy()
        run_steps(min(3, args.steps), upload=True)
        barrier(); dev_sync()
        tu = time.perf_counter()
        run_steps(args.steps, upload=True)
        for c in ctxs:
            c.synchronize()
        barrier()
        tu = time.perf_counter() - tu
        if world > 1:
            tm = torch.tensor([tu], dtype=torch.float64, device=cdev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            tu = float(tm.item())
        value_with_upload = world * B * args.steps / tu
    # the same kernels with one batch in flight, i.e. without other batches' kernels sharing the GPU, which is the
    # duration a kernel-level roofline should be read against
    isolated = {}
    if nctx > 1 and not stub:
        set_window(ctx, left.data_ptr(), right.data_ptr(), B)
        ctx.set_timing(2)   # one stream: the tracking chain would otherwise share the GPU with the disparity chain
        ctx.timing(reset=True)
        for _ in range(3):
            ctx.step(strict=not args.lenient)
        isolated = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in ctx.timing().items()}
        ctx.set_timing(1)
    # latency of one frame pair alone (batch of 1, one stream, nothing else in flight)
    latency_ms = None
    if not args.no_latency and not stub:
        set_window(ctx, left.data_ptr(), right.data_ptr(), 1)
        for _ in range(2):
            ctx.step()
        tl = time.perf_counter()
        for _ in range(10):
            ctx.step()
        latency_ms = (time.perf_counter() - tl) / 10 * 1e3

    if rank == 0:
        if strong:
            got = sum(a.shape[0] for a in gathered_poses.values())
            assert got == total_pairs, "gathered %d steps, expected %d" % (got, total_pairs)
        if args.dump_poses:
            np.savez(args.dump_poses, **{"seq%d" % k: v for k, v in gathered_poses.items()})
        steps = len(batches) if strong else args.steps
        fps = total_pairs / dt
        srows = stripe_rows(H)
        kb = kernel_bytes(W, H, D, srows, B)
        stage_ms = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in timing.items()}
        log("[stage ms per launch] " + ", ".join("%s=%.3f" % kv for kv in stage_ms.items()))
        # Kernel durations.  With several batches in flight the kernels of different batches share the GPU, so an event
        # pair around a launch measures a co-scheduled kernel (and rocprofv3, which serialises dispatches, would not
        # agree with it).  The roofline therefore uses the one-batch-in-flight pass made right after the timed region
        # (same process, same inputs, same HIP-event timers on the kernels' stream); the timed-region figures ride along.
        excl = isolated if isolated else stage_ms
        hbm_kernels = [k for k in kb if KERNEL_BOUND.get(k) == "hbm"]
        dom = max(hbm_kernels, key=lambda k: excl.get(k, 0.0))
        achieved = kb[dom] / (excl[dom] * 1e-3) / 1e9 if excl.get(dom, 0) > 0 else 0.0
        traffic = None
        traffic_per_frame = None   # HBM bytes of ALL the path's kernels per frame, from the same counter passes
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf) and args.workload in ("kitti", "kitti11", "4k"):
            try:
                tj = json.load(open(tf))
                wk = "4k" if args.workload == "4k" else "kitti"
                cb = tj.get(wk + "_batch", 8 if wk == "4k" else 16)   # the batch size the counters were collected at
                traffic = tj.get(wk, {}).get(dom)
                if traffic is not None:  # bytes scale with the frames of a launch
                    traffic = traffic * B / cb
                allk = tj.get(wk + "_all_kernels")   # every kernel of the path; older files hold the large kernels only
                traffic_per_frame = (allk if allk else sum(v for v in tj.get(wk, {}).values() if isinstance(v, (int, float)))) / cb
            except Exception:
                traffic = None

        # vector-instruction side of the same kernels (the tracker is bound by issue, not by HBM): share of the chip's
        # issue capacity = instructions x cycles per instruction / (SIMDs x clock x duration)
        valu = None
        vf = os.path.join(ROOT, "profiles", "valu.json")
        if os.path.exists(vf) and args.workload != "4k":
            try:
                vj = json.load(open(vf)).get("kitti", {})
                valu = {}
                for k, v in vj.items():
                    if isinstance(v, dict) and excl.get(k, 0) > 0:
                        insts = v["valu_insts_per_launch"] * B / vj.get("batch", 16)   # instructions scale with the frames of a launch
                        cyc = insts * v["issue_cycles_per_inst"] / vj["simds"]
                        valu[k] = {"valu_insts_per_launch": insts, "issue_cycles_per_inst": v["issue_cycles_per_inst"],
                                   "frac_of_issue_capacity": round(cyc / (vj["sclk_ghz"] * 1e9 * excl[k] * 1e-3), 3)}
            except Exception:
                valu = None

        def frac(ms, k):
            return round(kb[k] / (ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms.get(k, 0) > 0 else None

        whole_gbs = frame_bytes(W, H, D) * fps / world / 1e9
        config = {"workload": "%s: synthetic stereo, %dx%d, %d disparities, %d frame pairs per step per GPU, whole hot path "
                              "(SGBM-3way + FAST + circular KLT + triangulation + PnP-RANSAC)" % (args.workload, W, H, D, B),
                  "batch": B, "batches_in_flight": nctx, "hw_queues": int(os.environ["GPU_MAX_HW_QUEUES"]), "input_mode": args.mode,
                  "inputs": "uploaded from host memory every step" if args.upload else "resident in HBM",
                  "distinct_frames_per_rank": nframes, "windows_per_rank": nwin, "host_threads_per_rank": args.host_threads,
                  "pnp_minimal_solves": "gpu" if (not stub and p.pnp_device) else "host",
                  "result_gather": ("executed (%s, %d rank%s)" % ("RCCL" if backend == "nccl" else "gloo", world, "" if world == 1 else "s")) if collective else "skipped",
                  "sharding": ("11 KITTI 00-10-shaped sequences (%d steps), %s; relative poses all-gathered once (%s)" %
                               (total_pairs, "contiguous step ranges per rank" if args.chunked else "whole sequences per rank, longest first",
                                "RCCL" if backend == "nccl" else "gloo")) if strong else
                              "independent sequences, one per GPU; poses and the last batch's point clouds all-gathered once (%s)" %
                              ("RCCL" if backend == "nccl" else "gloo"),
                  "mpix_per_s": fps * W * H / 1e6, "latency_ms_one_pair_alone": latency_ms,
                  "value_with_upload": value_with_upload,
                  "features_per_frame": int(np.mean([i.n_features for i in info])) if len(info) else 0,
                  "matches_per_frame": int(np.mean([i.n_matched for i in info])) if len(info) else 0,
                  "stage_ms_timed_region": {k: round(v, 4) for k, v in stage_ms.items()},
                  "stage_ms_one_batch_in_flight": {k: round(v, 4) for k, v in excl.items()},
                  "whole_path_algorithmic_GBs": whole_gbs,
                  "rank_compute_s": [round(t, 4) for t in t_all]}
        if strong:
            config["steps_per_rank"] = loads
            config["imbalance_max_over_mean"] = round(max(loads) / (sum(loads) / len(loads)), 4) if sum(loads) else None
            config["sequence_lengths"] = lengths
        if stub:
            config["stub"] = True
        out = {
            "metric": METRIC,
            "value": fps, "unit": "stereo frames/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": dt / max(steps, 1) * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "s16 (disparity, FAST: u8/s16 integer; LK: int32/int64 accumulators + f32 solve; PnP: f64)",
            "data": "synthetic",
            "config": config,
        }
        if not stub:
            out["roofline"] = {
                "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel_choice": "longest launch among the kernels bound by HBM; the tracker (lk) is bound by vector-instruction issue, "
                                 "see valu_issue",
                # the whole path against the same roof: algorithmic bytes per frame x frames/s of one GPU
                # traffic_inflation: measured HBM bytes of all the path's kernels per frame (FETCH_SIZE x 2 + WRITE_SIZE counter
                # passes, profiles/traffic.json) / algorithmic bytes per frame
                "whole_path": {"algorithmic_GBs": whole_gbs, "frac": whole_gbs / HBM_PEAK_GBS,
                               "algorithmic_bytes_per_frame": frame_bytes(W, H, D),
                               "hbm_bytes_per_frame_measured": traffic_per_frame,
                               "traffic_inflation": (traffic_per_frame / frame_bytes(W, H, D)) if traffic_per_frame else None},
                # what plain streaming kernels reach on this GPU (tools/micro/hbm_rate.hip, measured once)
                "streaming_GBs_measured": {"read": 6290, "write": 3890, "copy": 4450, "1_read_4_writes": 4440},
                "algorithmic_bytes_per_launch": kb[dom], "avg_launch_ms": excl.get(dom, 0.0),
                "avg_launch_ms_timed_region": stage_ms.get(dom, 0.0), "frac_timed_region": frac(stage_ms, dom),
                "valu_issue": valu,
                "measured": "HIP events on the kernel's stream, one batch in flight, right after the timed region"
                            if isolated else "HIP events on the kernel's stream over the timed region",
                "per_kernel": {k: {"bound": KERNEL_BOUND.get(k), "algorithmic_GB_per_launch": round(kb[k] / 1e9, 4),
                                   "avg_launch_ms": round(excl.get(k, 0.0), 4), "frac": frac(excl, k),
                                   "avg_launch_ms_timed_region": round(stage_ms.get(k, 0.0), 4)} for k in kb}}
        if not args.no_cpu and world == 1 and not stub:   # the CPU baseline is a single-GPU-run item
            out["cpu_baseline"] = cpu_baseline(frames, seq, shape, args.cpu_pairs if args.cpu_pairs > 0 else (1 if args.workload == "4k" else 48))
        print(json.dumps(out), flush=True)
    if collective:
        with stdout_to_stderr():
            if world > 1:
                dist.barrier()
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
