"""Profiling aid: cycles per section (`make -C slam_amd/csrc stamps`) or loop trip counts (`make -C slam_amd/csrc
counts`) of the sector-query kernel; either target writes slam_amd/libmclhip_prof.so, which this script loads."""
import os, sys
os.environ["MCLHIP_LIBNAME"] = "libmclhip_prof.so"
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
z = ctx.raycast(advance_true_pose(X[0].copy(), (1.5, 0.0), 0.1), want_hits=False)[0][0]
base = [ctx.timing_get(100 + i)[0] for i in range(8)]
base8 = [ctx.timing_get(108 + i)[0] for i in range(8)]
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
