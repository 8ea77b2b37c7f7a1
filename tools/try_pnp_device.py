import sys, time
sys.path.insert(0, '.')
import numpy as np
from visualodometry_amd import abi, synth

def run(W, H, nb, dev, **kw):
    seq = synth.Sequence(seed=1000, width=W, height=H)
    frames = [seq.frame(i) for i in range(nb + 1)]
    p = abi.default_params(W, H, max_batch=nb, pnp_device=dev, **kw)
    ctx = abi.Context(p, 0)
    ctx.set_calib(seq.P0, seq.P1)
    left = np.stack([f[0] for f in frames]); right = np.stack([f[1] for f in frames])
    ctx.set_sequence(left, right)
    ctx.set_timing(True)
    Rt, info = ctx.step()
    t0 = time.perf_counter()
    for _ in range(3):
        Rt, info = ctx.step()
    dt = (time.perf_counter() - t0) / 3
    tm = ctx.timing() if hasattr(ctx, 'timing') else None
    ctx.close()
    return Rt, info, dt, tm

for (W, H, nb, kw) in [(1241, 376, 8, {}), (1241, 376, 8, dict(ransac_iters=1024, ransac_chunk=1024, max_features=20000, fast_threshold=10, fast_nms=0, lk_win=21))]:
    R0, i0, d0, t0 = run(W, H, nb, 0, **kw)
    R1, i1, d1, t1 = run(W, H, nb, 1, **kw)
    print("config", kw, "step ms host %.2f device %.2f" % (d0 * 1e3, d1 * 1e3))
    print(" max |Rt diff|", np.abs(R0 - R1).max())
    for b in range(nb):
        print("  b%d iters %d/%d inliers %d/%d ok %d/%d" % (b, i0[b].iters_run, i1[b].iters_run, i0[b].best_inliers, i1[b].best_inliers, i0[b].ok, i1[b].ok))
    print(" timing host", t0); print(" timing dev", t1)
