#!/usr/bin/env python3
"""Throughput benchmark of the stereo front-end hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--batch B] [--workload kitti|4k]

One "step" = one pass of the whole per-frame hot path (disparity, FAST, circular KLT, triangulation,
PnP-RANSAC; one iteration of the reference's main loop, Source/main.cpp:36-60) over a batch of B independent
synthetic stereo frame pairs that are already resident in HBM.  Rank r of an N-GPU run works on its own
sequence (seed 1000 + r): independent units, no collective inside the path; the relative poses are gathered
once at the end with RCCL.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def workload_shape(name):
    if name == "kitti":
        return dict(width=1241, height=376, num_disparities=96, lk_win=15, max_features=16384)
    if name == "4k":
        return dict(width=3840, height=2160, num_disparities=256, lk_win=15, max_features=65536)
    raise ValueError(name)


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
    vol_s = srows * w1 * D * 2      # one per-stripe-row s16 volume (cost, path sums)
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
        "sgbm_lrcheck": H * W * 4 + P * 2,
        "sgbm_median": P * 2 + P * 2 + P * 4,
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


def cpu_baseline(frames, seq, shape, sample_pairs):
    """The CPU oracle (a port of the same path) timed on the host cores on a bounded sample of the workload."""
    from oracle import binding as ob
    threads = max(1, min(16, int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))))
    ob.set_threads(threads)
    p = ob.default_params(shape["width"], shape["height"], num_disparities=shape["num_disparities"], lk_win=shape["lk_win"])
    pose = np.eye(4)
    npairs = sample_pairs
    nd = len(frames) - 1
    t0 = time.perf_counter()
    for k in range(npairs):
        i = k % nd
        ob.step(frames[i][0], frames[i][1], frames[i + 1][0], frames[i + 1][1], p, seq.P0, seq.P1, pose)
    dt = time.perf_counter() - t0
    # the same port on one thread, on a smaller sample
    ob.set_threads(1)
    n1 = max(1, min(3, npairs))
    t1 = time.perf_counter()
    for k in range(n1):
        i = k % nd
        ob.step(frames[i][0], frames[i][1], frames[i + 1][0], frames[i + 1][1], p, seq.P0, seq.P1, pose)
    dt1 = time.perf_counter() - t1
    ob.set_threads(threads)
    return {"value": npairs / dt, "unit": "stereo frames/s", "cores": threads, "kind": "port",
            "sample": "%d frame pairs of the same synthetic %dx%d sequence, %.1f s wall (SGBM on 4 stripe threads, LK on %d)" %
                      (npairs, shape["width"], shape["height"], dt, threads),
            "single_thread_value": n1 / dt1, "single_thread_sample": "%d frame pairs, %.1f s wall" % (n1, dt1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0, help="frame pairs per step (default: 16 at KITTI shape, 8 at 4K)")
    ap.add_argument("--workload", default="kitti", choices=["kitti", "4k"])
    ap.add_argument("--distinct", type=int, default=4, help="distinct frame pairs rendered (tiled to fill the batch)")
    ap.add_argument("--cpu-pairs", type=int, default=12)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--inflight", type=int, default=0,
                    help="contexts taking turns (overlaps host PnP with GPU work; default: 3 at KITTI shape, 2 at 4K, "
                         "where a context holds 5.5 GB per frame pair)")
    ap.add_argument("--upload", action="store_true",
                    help="re-upload the batch's images from host memory before every step (PCIe-inclusive rate; the headline "
                         "value is measured without it, inputs resident in HBM)")
    ap.add_argument("--latency", action="store_true",
                    help="also time one frame pair alone (batch of 1) after the run; off by default so that a kernel profile "
                         "of the default command holds launches of one batch size only")
    ap.add_argument("--mode", default="pairs", choices=["pairs", "sequence"],
                    help="pairs: B independent frame pairs, 4 pyramids each (what the reference rebuilds per step); "
                         "sequence: B + 1 consecutive frames, every frame pyramided once")
    args = ap.parse_args()
    if args.batch <= 0:
        args.batch = 8 if args.workload == "4k" else 16
    if args.inflight <= 0:
        args.inflight = 2 if args.workload == "4k" else 3

    import torch
    import torch.distributed as dist
    from visualodometry_amd import abi, synth

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if "VO_BENCH_DEVICE" in os.environ:  # rehearsal of the multi-rank path on a box with fewer GPUs than ranks
        local_rank = int(os.environ["VO_BENCH_DEVICE"])
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL ("nccl") is the backend of a real run; gloo only for rehearsing the multi-rank flow on one GPU
        backend = os.environ.get("VO_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    shape = workload_shape(args.workload)
    W, H, D = shape["width"], shape["height"], shape["num_disparities"]
    B = args.batch
    seq = synth.Sequence(seed=1000 + rank, width=W, height=H)
    nd = max(1, min(args.distinct, B))
    t0 = time.perf_counter()
    frames = [seq.frame(i) for i in range(nd + 1)]
    log("[rank %d] rendered %d synthetic frames in %.1f s" % (rank, nd + 1, time.perf_counter() - t0))

    def tiled(which, shift):
        arr = np.stack([frames[(b % nd) + shift][which] for b in range(B)])
        return torch.from_numpy(arr).to(dev)

    if args.mode == "pairs":
        lt1, rt1, lt2, rt2 = tiled(0, 0), tiled(1, 0), tiled(0, 1), tiled(1, 1)  # resident in HBM before timing starts
    else:
        # B + 1 consecutive frames: walk the rendered frames forth and back so that every pair is a real motion
        walk = [m if m <= nd else 2 * nd - m for m in (k % (2 * nd) for k in range(B + 1))]
        left = torch.from_numpy(np.stack([frames[w][0] for w in walk])).to(dev)
        right = torch.from_numpy(np.stack([frames[w][1] for w in walk])).to(dev)

    p = abi.default_params(W, H, max_batch=B, num_disparities=D, lk_win=shape["lk_win"], max_features=shape["max_features"],
                           host_threads=args.host_threads)
    # `inflight` contexts take turns: while the host part (minimal solvers, refine) of one batch runs, the GPU
    # part of the next batch is already enqueued on the other context's stream
    ctxs = []
    for _ in range(max(1, args.inflight)):
        c = abi.Context(p, local_rank)
        c.set_calib(seq.P0, seq.P1)
        if args.mode == "pairs":
            c.set_images_device(B, [t.data_ptr() for t in (lt1, rt1, lt2, rt2)], W, W * H, keep=(lt1, rt1, lt2, rt2))
        else:
            c.set_sequence_device(B, left.data_ptr(), right.data_ptr(), W, W * H, keep=(left, right))
        ctxs.append(c)
    ctx = ctxs[0]
    nctx = len(ctxs)

    host_imgs = None
    if args.upload:
        if args.mode == "pairs":
            host_imgs = [t.cpu().numpy() for t in (lt1, rt1, lt2, rt2)]
        else:
            host_imgs = [left.cpu().numpy(), right.cpu().numpy()]

    def submit(c):
        if host_imgs is not None:   # host -> device copy of the whole batch, then the step
            if args.mode == "pairs":
                c.set_images(*host_imgs)
            else:
                c.set_sequence(*host_imgs)
        c.step_submit()

    def run_steps(k):
        """k whole batches through the path, pipeline fill and drain included"""
        out = None
        for s in range(min(nctx, k)):
            submit(ctxs[s])
        for s in range(k):
            c = ctxs[s % nctx]
            out = c.step_finish()
            if s + nctx < k:
                submit(c)
        return out

    if args.warmup > 0:
        Rt, info = run_steps(args.warmup)
    for c in ctxs:
        c.set_timing(True)
        c.timing(reset=True)

    def barrier():
        if world > 1:
            dist.barrier()

    barrier(); torch.cuda.synchronize()
    for c in ctxs:
        c.synchronize()
    t0 = time.perf_counter()
    Rt, info = run_steps(args.steps)
    for c in ctxs:
        c.synchronize()
    torch.cuda.synchronize()
    # the only collective: gather the per-frame relative poses of every rank (result gather, RCCL over xGMI)
    cdev = dev if (world == 1 or dist.get_backend() == "nccl") else torch.device("cpu")
    poses = torch.from_numpy(Rt).to(cdev)
    if world > 1:
        gathered = [torch.empty_like(poses) for _ in range(world)]
        dist.all_gather(gathered, poses)
        # ... and the point clouds (triangulated points) of the last batch, as fixed-size padded buffers
        from visualodometry_amd import sequence as vseq
        last = ctxs[(args.steps - 1) % nctx]
        clouds = vseq.gather_point_clouds([last.object_points(b)[0] for b in range(B)], cap=shape["max_features"])
        assert len(clouds) == world
        torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    timing = {}
    for c in ctxs:
        for k, v in c.timing().items():
            a = timing.get(k, (0.0, 0))
            timing[k] = (a[0] + v[0], a[1] + v[1])
    # outside the timed region: the same kernels with one batch in flight, i.e. without other batches' kernels
    # sharing the GPU, which is the duration a kernel-level roofline should be read against
    isolated = {}
    if nctx > 1:
        ctx.timing(reset=True)
        for _ in range(3):
            ctx.step()
        isolated = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in ctx.timing().items()}
    # latency of one frame pair alone (batch of 1, one stream, nothing else in flight), outside the timed region
    latency_ms = None
    if args.latency and args.mode == "pairs":
        ctx.set_images_device(1, [t.data_ptr() for t in (lt1, rt1, lt2, rt2)], W, W * H, keep=(lt1, rt1, lt2, rt2))
        for _ in range(2):
            ctx.step()
        tl = time.perf_counter()
        for _ in range(10):
            ctx.step()
        latency_ms = (time.perf_counter() - tl) / 10 * 1e3
    if rank == 0:
        steps = args.steps
        fps = world * B * steps / dt
        srows = stripe_rows(H)
        kb = kernel_bytes(W, H, D, srows, B)
        stage_ms = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in timing.items()}
        log("[stage ms per launch] " + ", ".join("%s=%.3f" % kv for kv in stage_ms.items()))
        # Kernel durations.  With several batches in flight the kernels of different batches share the GPU, so an event
        # pair around a launch measures a co-scheduled kernel (and rocprofv3, which serialises dispatches, would not
        # agree with it).  The roofline therefore uses the one-batch-in-flight pass made right after the timed region
        # (same process, same inputs, same HIP-event timers on the kernels' stream); the timed-region figures ride along.
        excl = isolated if isolated else stage_ms
        dom = max(kb, key=lambda k: excl.get(k, 0.0))
        achieved = kb[dom] / (excl[dom] * 1e-3) / 1e9 if excl.get(dom, 0) > 0 else 0.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(args.workload, {}).get(dom)
            except Exception:
                traffic = None

        # vector-instruction side of the same kernels (the tracker is bound by issue, not by HBM): share of the chip's
        # issue capacity = instructions x cycles per instruction / (SIMDs x clock x duration)
        valu = None
        vf = os.path.join(ROOT, "profiles", "valu.json")
        if os.path.exists(vf):
            try:
                vj = json.load(open(vf)).get(args.workload, {})
                valu = {}
                for k, v in vj.items():
                    if isinstance(v, dict) and excl.get(k, 0) > 0:
                        cyc = v["valu_insts_per_launch"] * v["issue_cycles_per_inst"] / vj["simds"]
                        valu[k] = {"valu_insts_per_launch": v["valu_insts_per_launch"], "issue_cycles_per_inst": v["issue_cycles_per_inst"],
                                   "frac_of_issue_capacity": round(cyc / (vj["sclk_ghz"] * 1e9 * excl[k] * 1e-3), 3)}
            except Exception:
                valu = None

        def frac(ms, k):
            return round(kb[k] / (ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms.get(k, 0) > 0 else None

        out = {
            "metric": "stereo frames/sec at 1241x376 (1/2/4/8 GPU) + achieved HBM GB/s vs roofline",
            "value": fps, "unit": "stereo frames/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "s16 (disparity, FAST: u8/s16 integer; LK: int32/int64 accumulators + f32 solve; PnP: f64)",
            "data": "synthetic",
            "config": {"workload": "%s-shape synthetic stereo, %dx%d, %d disparities, %d frame pairs per step per GPU, whole hot path "
                                   "(SGBM-3way + FAST + circular KLT + triangulation + PnP-RANSAC)" % (args.workload, W, H, D, B),
                       "batch": B, "batches_in_flight": nctx, "input_mode": args.mode, "inputs": "uploaded from host memory every step" if args.upload else "resident in HBM", "sharding": "independent sequences, one per GPU; poses and the last batch's point clouds all-gathered once (RCCL)",
                       "mpix_per_s": fps * W * H / 1e6, "latency_ms_one_pair_alone": latency_ms,
                       "features_per_frame": int(np.mean([i.n_features for i in info])),
                       "matches_per_frame": int(np.mean([i.n_matched for i in info])),
                       "stage_ms_timed_region": {k: round(v, 4) for k, v in stage_ms.items()},
                       "stage_ms_one_batch_in_flight": {k: round(v, 4) for k, v in excl.items()},
                       "whole_path_algorithmic_GBs": frame_bytes(W, H, D) * fps / world / 1e9},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         # what plain streaming kernels reach on this GPU (tools/micro/hbm_rate.hip, measured once)
                         "streaming_GBs_measured": {"read": 6290, "write": 3890, "copy": 4450, "1_read_4_writes": 4440},
                         "algorithmic_bytes_per_launch": kb[dom], "avg_launch_ms": excl.get(dom, 0.0),
                         "avg_launch_ms_timed_region": stage_ms.get(dom, 0.0), "frac_timed_region": frac(stage_ms, dom),
                         "valu_issue": valu,
                         "measured": "HIP events on the kernel's stream, one batch in flight, right after the timed region"
                                     if isolated else "HIP events on the kernel's stream over the timed region",
                         # the tracker and the resolve pass are bound by vector instruction issue, not by HBM: their
                         # fractions say how far below the memory roof they sit
                         "per_kernel": {k: {"algorithmic_GB_per_launch": round(kb[k] / 1e9, 4), "avg_launch_ms": round(excl.get(k, 0.0), 4),
                                            "frac": frac(excl, k), "avg_launch_ms_timed_region": round(stage_ms.get(k, 0.0), 4)} for k in kb}},
        }
        if not args.no_cpu and world == 1:   # the CPU baseline is a single-GPU-run item
            out["cpu_baseline"] = cpu_baseline(frames, seq, shape, args.cpu_pairs)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
