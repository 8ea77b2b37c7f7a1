"""Latency of one frame pair alone (one context, batch of 1) against the stream schedule and the hardware queue count.
Usage: python tools/try_latency.py QUEUES [single]"""
import os, sys, time
q = sys.argv[1]
os.environ["GPU_MAX_HW_QUEUES"] = q
if len(sys.argv) > 2 and sys.argv[2] == "single":
    os.environ["VO_SINGLE_STREAM"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from visualodometry_amd import abi, synth
W, H = 1241, 376
seq = synth.Sequence(seed=1000, width=W, height=H)
f = [seq.frame(i) for i in range(2)]
nctx = int(os.environ.get("NCTX", "1"))
ctxs = [abi.Context(abi.default_params(W, H, max_batch=42 if nctx > 1 else 1), 0) for _ in range(nctx)]
ctx = ctxs[0]
for c in ctxs:
    c.set_calib(seq.P0, seq.P1)
    c.set_sequence(np.stack([f[0][0], f[1][0]]), np.stack([f[0][1], f[1][1]]))
    c.step()
for _ in range(5):
    ctx.step()
t0 = time.perf_counter()
for _ in range(50):
    ctx.step()
print("queues %s %s nctx %d: %.3f ms per step of one pair" % (q, "single stream" if "VO_SINGLE_STREAM" in os.environ else "two streams", nctx, (time.perf_counter() - t0) / 50 * 1e3))
