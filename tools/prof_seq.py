"""Host-side timing of one context on sequence input: upload (vo_set_sequence) and step, then the per-stage timers.
    python tools/prof_seq.py      (needs a GPU)"""
import time, numpy as np, sys
sys.path.insert(0, '.')
from visualodometry_amd import abi, synth
seq = synth.Sequence(seed=1000)
fr = [seq.frame(i) for i in range(3)]
left = np.stack([fr[i % 3][0] for i in range(17)]); right = np.stack([fr[i % 3][1] for i in range(17)])
ctx = abi.Context(abi.default_params(1241, 376, max_batch=16)); ctx.set_calib(seq.P0, seq.P1)
for it in range(4):
    t0 = time.perf_counter(); ctx.set_sequence(left, right); ctx.synchronize(); t1 = time.perf_counter()
    Rt, info = ctx.step(); t2 = time.perf_counter()
    print("set_sequence %.1f ms, step %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
ctx.set_timing(True)
ctx.step(); 
print({k: round(v[0], 2) for k, v in ctx.timing().items()})
