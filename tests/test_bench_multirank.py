"""bench.py's multi-rank path end to end on CPU ranks (gloo): `--gpus 2` without a launcher makes bench.py start the two
ranks itself; a stand-in context (no kernels, 'poses' = a known function of the frames it was handed) lets the test check
that each rank processed the frame pairs it was assigned and that the one gather returns them in order."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from visualodometry_amd import sequence, synth  # noqa: E402

W, H, B, NF = 64, 32, 4, 9


def run_bench(extra, tmp_path):
    dump = str(tmp_path / "poses.npz")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--stub-context", "--no-cpu", "--width", str(W), "--height", str(H),
           "--batch", str(B), "--frames", str(NF), "--dump-poses", dump] + extra
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, timeout=300, check=True).stdout.decode()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0]), np.load(dump)


def frame_sums(seed):
    s = synth.Sequence(seed=seed, width=W, height=H)
    return [int(s.frame(i)[0].astype(np.int64).sum()) for i in range(NF)]


def expected_kitti11(scale):
    lengths = [max(2, int(round(n * scale))) for n in bench.KITTI_SEQ_LENGTHS]
    sums = frame_sums(1000)
    period = 2 * (NF - 1)

    def idx(i, t):
        m = (t + 7 * i) % period
        return m if m <= NF - 1 else period - m
    want = {}
    for i, n in enumerate(lengths):
        want[i] = np.stack([bench.StubContext.pose_of(sums[idx(i, t)], sums[idx(i, t + 1)]).reshape(16) for t in range(n - 1)])
    return lengths, want


def test_gpus2_self_launch_kitti11_whole_sequences(tmp_path):
    scale = 0.004
    j, poses = run_bench(["--gpus", "2", "--workload", "kitti11", "--scale", str(scale), "--inflight", "2", "--warmup", "1"], tmp_path)
    lengths, want = expected_kitti11(scale)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["stub"] is True
    assert j["config"]["sequence_lengths"] == lengths
    owner = sequence.assign_sequences(lengths, 2)
    assert j["config"]["steps_per_rank"] == [sum(lengths[i] - 1 for i in o) for o in owner]
    total = sum(n - 1 for n in lengths)
    assert abs(j["value"] * j["ms_per_step"] * 1e-3 * j["steps"] - total) < 1e-6 * total   # value = all steps / max-rank time
    for i in range(len(lengths)):
        assert np.array_equal(poses["seq%d" % i], want[i]), i


def test_gpus2_self_launch_kitti11_chunked(tmp_path):
    scale = 0.004
    j, poses = run_bench(["--gpus", "2", "--workload", "kitti11", "--chunked", "--scale", str(scale), "--inflight", "3", "--warmup", "0"], tmp_path)
    lengths, want = expected_kitti11(scale)
    loads = j["config"]["steps_per_rank"]
    assert j["n_gpus"] == 2 and max(loads) - min(loads) <= 1 and sum(loads) == sum(n - 1 for n in lengths)
    for i in range(len(lengths)):
        assert np.array_equal(poses["seq%d" % i], want[i]), i


def test_gpus2_self_launch_weak_scaling_default_workload(tmp_path):
    steps = 3
    j, poses = run_bench(["--gpus", "2", "--steps", str(steps), "--warmup", "1", "--inflight", "2"], tmp_path)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["steps"] == steps
    assert abs(j["value"] - 2 * B * steps / (j["ms_per_step"] * 1e-3 * steps)) < 1e-6 * j["value"]
    nwin = (NF - 1) // B
    f0 = ((steps - 1) % nwin) * B            # the window the last step of every rank worked on
    for r in range(2):
        sums = frame_sums(1000 + r)          # rank r streams its own sequence
        want = np.stack([bench.StubContext.pose_of(sums[f0 + b], sums[f0 + b + 1]) for b in range(B)])
        assert np.array_equal(poses["seq%d" % r], want), r


def test_one_rank_needs_no_process_group(tmp_path):
    j, poses = run_bench(["--steps", "2", "--warmup", "0", "--inflight", "1"], tmp_path)
    assert j["n_gpus"] == 1 and poses["seq0"].shape == (B, 4, 4)
