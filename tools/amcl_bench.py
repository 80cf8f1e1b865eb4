"""Timing of augmented MCL with every random number drawn on the device (2^20 particles x the reference's 13 beams
on the synthetic 4096^2 map by default)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib as L, workload as W
from slam_amd.mc_localization import advance_true_pose
M = int(os.environ.get("AMCL_M", 1 << 20)); steps = int(os.environ.get("AMCL_STEPS", 12)); B = int(os.environ.get("AMCL_B", 13))
angles = [0, 15, 30, 45, 60, 75, 90, 345, 330, 315, 300, 285, 270] if B == 13 else list(W.beam_angles(B))
grid = W.synthetic_map(4096)
ext = 4096 * W.RES
ctx = L.Context(0, L.F32)
ctx.set_map(grid, W.RES, W.RES)
w4 = np.array([0.97, 0.005, 0.020, 0.005]); w4 /= w4.sum()
ctx.set_sensor(angles, W.Z_MAX, 0.15, 1.0, w4, 5 * np.pi / 180.0)
ctx.init_particles(M, None, ext, ext, 5)
X0 = ctx.get_particles()[0]
x_gt = X0[0].copy()
w_slow = w_fast = 0.0
ts, inj = [], []
for t in range(steps):
    ut = (1.5, 0.0) if t % 2 == 0 else (1.5, 0.5)
    x_gt = advance_true_pose(x_gt, ut, 0.1)
    z = ctx.raycast(x_gt, want_hits=False)[0][0]
    t0 = time.perf_counter()
    w_avg = ctx.sample_augmented_device(ut[0], ut[1], 0.1, [4, 2, 4, 4, 4, 2], z, 1000 + t, 0.15)
    w_slow += 0.001 * (w_avg - w_slow); w_fast += 0.9 * (w_avg - w_fast)          # mc_localization2.py:248-249
    p_random = max(0.0, 1 - w_fast / w_slow) if w_slow > 0 else 0.0
    ctx.normalise(L.W_PRODUCT)
    n_inj = ctx.resample_augmented_device(0.37 / M, p_random, ext, ext, 2000 + t)
    ts.append(time.perf_counter() - t0); inj.append(n_inj)
    print("step %d %.2f ms  w_avg %.3e  p_random %.3f  injected %d" % (t, ts[-1] * 1e3, w_avg, p_random, n_inj), flush=True)
ms = float(np.median(ts[2:])) * 1e3
print(json.dumps({"metric": "augmented MCL particle-beam evaluations per second (device-drawn noise and injections)",
                  "particles": M, "beams": B, "ms_per_step": ms, "value": M * B / (ms * 1e-3), "unit": "particle-beam evals/s",
                  "particle_steps_per_s": M / (ms * 1e-3), "injected_per_step": inj}))
