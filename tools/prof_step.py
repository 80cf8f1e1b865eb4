"""Profiling driver: a few fused steps of the benchmark workload (no CPU baseline, no timing spans)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib, workload as W
from slam_amd.mc_localization import advance_true_pose
M = int(os.environ.get("MCL_M", 1 << 20)); B = int(os.environ.get("MCL_B", 180)); N = int(os.environ.get("MCL_MAP", 4096))
steps = int(os.environ.get("MCL_STEPS", 3))
grid = W.synthetic_map(N, target_occupancy=float(os.environ.get("MCL_OCC", 0.04)))
ctx = _lib.Context(0, _lib.F32 if os.environ.get("MCL_DTYPE", "f32") == "f32" else _lib.F64)
ctx.set_map(grid, W.RES, W.RES)
w4 = np.array([0.97, 0.005, 0.020, 0.005]); w4 /= w4.sum()
ctx.set_sensor(W.beam_angles(B), W.Z_MAX, 0.15, 1.0, w4, 5 * np.pi / 180.0)
X = W.uniform_particles(grid, M)
ctx.set_particles(X)
ctx.snapshot_particles()
x_gt = advance_true_pose(X[0].copy(), (1.5, 0.0), 0.1)
z = ctx.raycast(x_gt, want_hits=False)[0][0]
for t in range(steps):
    ctx.restore_particles_async()
    ctx.step(1.5, 0.0, 0.1, [4, 2, 4, 4, 4, 2], z, 0.3 / M, None, 7 + t, _lib.W_LOG, wait=False)
ctx.sync()
if os.environ.get("MCL_CELLS"):
    # mean number of obstacle cells within z_max of a pose (the cells the sector query has to look at), on a sample
    from scipy.spatial import cKDTree
    oi, oj = np.nonzero(grid == 0)
    tree = cKDTree(np.stack([oj * np.float32(W.RES), oi * np.float32(W.RES)], 1).astype(np.float64))
    n = tree.query_ball_point(X[:4096, :2], W.Z_MAX, return_length=True)
    print("cells_in_range %.3f occupancy %.5f beams %d" % (float(np.mean(n)), float(np.mean(grid == 0)), B))
print("done")
