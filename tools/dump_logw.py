"""Debug helper: one sample() of the benchmark workload, log-weights saved to gpurun_out/."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib, workload as W
from slam_amd.mc_localization import advance_true_pose
M, B = 1 << 20, 180
grid = W.synthetic_map(4096)
ctx = _lib.Context(0, _lib.F32)
ctx.set_map(grid, W.RES, W.RES)
w4 = np.array([0.97, 0.005, 0.020, 0.005]); w4 /= w4.sum()
ctx.set_sensor(W.beam_angles(B), W.Z_MAX, 0.15, 1.0, w4, 5 * np.pi / 180.0)
X = W.uniform_particles(grid, M)
ctx.set_particles(X)
x_gt = advance_true_pose(X[0].copy(), (1.5, 0.0), 0.1)
z = ctx.raycast(x_gt, want_hits=False)[0][0]
ctx.sample(1.5, 0.0, 0.1, [4, 2, 4, 4, 4, 2], z, None, seed=5)
_, w, lw = ctx.get_particles(False, True, True)
np.save("gpurun_out/logw.npy", lw.astype(np.float64))
print("max", lw.max(), "finite", np.isfinite(lw).sum(), "median", np.median(lw[np.isfinite(lw)]))
