"""Timing of FastSLAM 1.0 on the large configuration (2^20 particles x 256 landmarks by default)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib as L
from slam_amd import fast_slam1 as S
M = int(os.environ.get("FS_M", 1 << 20)); NL = int(os.environ.get("FS_L", 256)); steps = int(os.environ.get("FS_STEPS", 6))
rng = np.random.RandomState(9)
# landmarks scattered over the area the reference's trajectory covers, ~25 % of them within r_max at a time
lm = np.stack([rng.uniform(-8, 33, NL), rng.uniform(-24, 8, NL), np.zeros(NL)], 1).astype(np.float32)
slam = S.FastSLAM(M=M, landmarks=lm, capacity=NL + 64, noise="device")
t_all = []
for t in range(steps):
    ut = [25, 0.001] if t % 5 else [25, -np.pi / 2 * 5]
    t0 = time.perf_counter()
    slam.control(ut)
    t_all.append(time.perf_counter() - t0)
    print("step", t, "%.1f ms" % (t_all[-1] * 1e3), flush=True)
st_n = slam._ctx.fs_get_state()["nfeat"] if M <= (1 << 18) else None
ms = float(np.median(t_all[1:])) * 1e3
print({"particles": M, "landmarks": NL, "ms_per_step_median": ms, "particle_steps_per_s": M / (ms * 1e-3),
       "mean_features": None if st_n is None else float(st_n.mean())})
