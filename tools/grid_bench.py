"""Timing of the histogram filter on the large configuration (2048^2 cells x 72 headings by default)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib as L, workload as W
N = int(os.environ.get("GRID_N", 2048)); D = int(os.environ.get("GRID_D", 72)); steps = int(os.environ.get("GRID_STEPS", 2))
angles = [0, 15, 30, 45, 60, 75, 90, 345, 330, 315, 300, 285, 270]
grid = W.synthetic_map(N)
ctx = L.Context(0, L.F64)
t0 = time.time()
ctx.grid_set_map(grid, W.RES, W.RES, 200)
w4 = np.array([0.97, 0.005, 0.020, 0.005]); w4 /= w4.sum()
ctx.grid_set_sensor(angles, D, 360.0 / D, W.Z_MAX, 0.15, 1.0, w4, 5 * np.pi / 180.0)
ctx.grid_init_belief()
print("set-up %.1f s" % (time.time() - t0), flush=True)
z = np.array([2.0, 3.1, 5.0, 1.2, 0.8, 4.4, 5.0, 2.2, 2.9, 3.3, 5.0, 1.0, 0.6])
ctx.timing_enable(True)
for t in range(steps):
    ctx.grid_predict(1.5, 0.0 if t % 2 == 0 else 0.5, 0.1, [0.2, 0.1, 0.2, 0.2, 0.1, 0.1])
    s = ctx.grid_update(z)
    print("step", t, "sum", s, flush=True)
np_, msp = ctx.timing_get(3); nu, msu = ctx.timing_get(4)
cells = N * N
print({"grid": [N, N, D], "predict_ms": msp / max(np_, 1), "update_ms": msu / max(nu, 1),
       "predict_GFLOPs": 2.0 * cells * D * D * 49 / (msp / max(np_, 1) * 1e-3) / 1e9,
       "update_cell_heading_beams_per_s": cells * D * 13 / (msu / max(nu, 1) * 1e-3)})
