"""Cost of the sector query as a function of the particles' heading: every particle gets the same heading (plus a
spread of 0.01 rad), no motion noise, and the fused step is timed.  Shows what the headings outside the circular range
of the interval path cost (MCL_TH, default a sweep)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib, workload as W
M, B = 1 << 20, 180
grid = W.synthetic_map(4096)
ctx = _lib.Context(0, _lib.F32)
ctx.set_map(grid, W.RES, W.RES)
w4 = np.array([0.97, 0.005, 0.020, 0.005]); w4 /= w4.sum()
ctx.set_sensor(W.beam_angles(B), W.Z_MAX, 0.15, 1.0, w4, 5 * np.pi / 180.0)
X = W.uniform_particles(grid, M)
z = ctx.raycast(X[0].copy(), want_hits=False)[0][0]
rng = np.random.RandomState(1)
for th in [float(v) for v in os.environ.get("MCL_TH", "3.0 -3.0 6.2 -6.2 8.0 12.5 20.0").split()]:
    X[:, 2] = th + rng.uniform(-0.005, 0.005, M)
    ctx.set_particles(X)
    ctx.snapshot_particles()
    ts = []
    for t in range(8):
        ctx.restore_particles_async()
        ctx.sync()
        t0 = time.perf_counter()
        ctx.step(0.0, 1e-12, 0.1, [0, 0, 0, 0, 0, 0], z, 0.3 / M, None, 7 + t, _lib.W_LOG, wait=True)
        ts.append(time.perf_counter() - t0)
    print("heading %6.2f: %.3f ms per step" % (th, 1e3 * float(np.median(ts[2:]))))
