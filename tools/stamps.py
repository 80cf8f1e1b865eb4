"""Profiling aid: cycles per section (`make -C slam_amd/csrc stamps`) or loop trip counts (`make -C slam_amd/csrc
counts`) of the sector-query kernel; either target writes slam_amd/libmclhip_prof.so, which this script loads."""
import os, sys
os.environ.setdefault("MCLHIP_LIBNAME", "libmclhip_prof.so")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib, workload as W
from slam_amd.mc_localization import advance_true_pose
GRID_MODE = len(sys.argv) > 1 and sys.argv[1] == "grid"     # the histogram filter's update instead of the particle filter's
M, B = 1 << 20, 180
grid = W.synthetic_map(1024 if GRID_MODE else 4096)
ctx = _lib.Context(0, _lib.F64 if GRID_MODE else _lib.F32)
w4 = np.array([0.97, 0.005, 0.020, 0.005]); w4 /= w4.sum()
if GRID_MODE:
    ctx.grid_set_map(grid, W.RES, W.RES, 200)
    ctx.grid_set_sensor([0, 15, 30, 45, 60, 75, 90, 345, 330, 315, 300, 285, 270], 72, 5.0, W.Z_MAX, 0.15, 1.0, w4, 5 * np.pi / 180.0)
    ctx.grid_init_belief()
    ctx.grid_predict(1.5, 0.0, 0.1, [0.2, 0.1, 0.2, 0.2, 0.1, 0.1])
    M = int(np.count_nonzero(grid >= 200))      # one wavefront per free cell
else:
    ctx.set_map(grid, W.RES, W.RES)
    ctx.set_sensor(W.beam_angles(B), W.Z_MAX, 0.15, 1.0, w4, 5 * np.pi / 180.0)
    X = W.uniform_particles(grid, M)
    ctx.set_particles(X)
    z = ctx.raycast(advance_true_pose(X[0].copy(), (1.5, 0.0), 0.1), want_hits=False)[0][0]
base = [ctx.timing_get(100 + i)[0] for i in range(8)]
base8 = [ctx.timing_get(108 + i)[0] for i in range(8)]
if GRID_MODE:
    ctx.grid_update(np.array([2.0, 3.1, 5.0, 1.2, 0.8, 4.4, 5.0, 2.2, 2.9, 3.3, 5.0, 1.0, 0.6]))
else:
    ctx.step(1.5, 0.0, 0.1, [4, 2, 4, 4, 4, 2], z, 0.3 / M, None, 7, _lib.W_LOG, wait=True)
now = [ctx.timing_get(100 + i)[0] for i in range(8)]
d = np.array(now, float) - np.array(base, float)
names = ["setup", "tile lists+scan", "fetch", "prefilter+ring", "match: geometry/bearing/window", "match: beam groups", "likelihood+store", "loop top (stamp overhead)"]
for n, v in zip(names, d):
    print("%-34s %6.1f %%   %8.0f cycles/particle" % (n, 100 * v / d.sum(), v / M))

cn = ["batches (wave)", "match calls (wave)", "cells within z_max (lane)", "window passes (wave)", "group steps (wave)",
      "group steps (lane)", "edge branches (wave)", "minimum updates (lane)"]
c = np.array([ctx.timing_get(108 + i)[0] for i in range(8)], float) - np.array(base8, float)
for n, v in zip(cn, c):
    print("%-34s %10.3f per particle" % (n, v / M))
if os.environ.get("MCL_COUNTS_JSON") and not GRID_MODE:
    # (counting build) what the cell loop of the interval-only kernel fetched and what of it was within z_max
    import json
    from scipy.spatial import cKDTree
    oi, oj = np.nonzero(grid == 0)
    tree = cKDTree(np.stack([oj * np.float32(W.RES), oi * np.float32(W.RES)], 1).astype(np.float64))
    cells = float(np.mean(tree.query_ball_point(X[:8192, :2], W.Z_MAX, return_length=True)))
    batches = float(c[0]) / M
    json.dump({"batches_of_64_per_particle": batches, "lane_slots_fetched_per_particle": 64.0 * batches,
               "cells_within_z_max_per_particle": cells,
               "lane_utilisation_of_the_cell_loop": cells / (64.0 * batches),
               "note": "counting build (make -C slam_amd/csrc counts), one step of the benchmark workload; both launches of the query"},
              open(os.environ["MCL_COUNTS_JSON"], "w"), indent=1)
