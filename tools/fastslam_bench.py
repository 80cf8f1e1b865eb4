"""Timing of FastSLAM 1.0 on the large configuration (2^20 particles x 256 landmarks by default)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from slam_amd import _lib as L
from slam_amd import fast_slam1 as S
M = int(os.environ.get("FS_M", 1 << 20)); NL = int(os.environ.get("FS_L", 256)); steps = int(os.environ.get("FS_STEPS", 6))
np.random.seed(123)                 # the resampler's uniform draws come from np.random, as in the reference
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
nfeat, _ = slam._ctx.fs_feature_counts()
mean_features = float(nfeat.mean())
ms = float(np.median(t_all[1:])) * 1e3
# The data movement of the reference's step: every live feature slot (64 B) is read and written once by the particle
# update and once more by the resampling copy.  (This implementation moves less: weights first, updates only for the
# drawn particles -- see `traffic`, the bytes the kernels of one step actually exchanged with HBM in the profiled run.)
algo_bytes = M * mean_features * 64.0 * 4
from profile_facts import latest, traffic_per_step
src, pmc = latest("fastslam_pmc_per_kernel")
traffic, per_kernel = (None, None)
if pmc is not None and (M, NL) == (1 << 20, 256):
    n_steps = max(c["FETCH_SIZE"]["launches"] for k, c in pmc.items() if "fs_associate" in k and "FETCH_SIZE" in c)
    # (fs_unshare_kernel belongs to the read-out of the state after the run, not to a step)
    traffic, per_kernel = traffic_per_step(pmc, n_steps, keep=lambda name: "fs_unshare" not in name)
out = {"metric": "FastSLAM 1.0 particle-steps per second", "particles": M, "landmarks": NL, "value": M / (ms * 1e-3),
       "unit": "particle-steps/s", "ms_per_step": ms, "mean_features": mean_features,
       "roofline": {"bound": "hbm", "kernel": "whole step (pose, draw + associate, weigh, resample, update + copy of the drawn)",
                    "achieved": algo_bytes / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                    "frac": algo_bytes / (ms * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_step": algo_bytes,
                    "traffic": traffic, "traffic_per_kernel": per_kernel, "source": src,
                    "measured_hbm_frac": None if traffic is None else traffic / (ms * 1e-3) / 8e12}}
if os.environ.get("FS_CPU", "1") == "1":
    # CPU baseline: the numpy restatement on 64 particles of the same landmark field (one core)
    from oracle import fastslam_oracle as F
    orc = F.OracleFastSLAM(64, lm)
    rs = np.random.RandomState(1)
    tt = []
    for t in range(4):
        ut = [25, 0.001]
        nv = sum(1 for a in F.true_measurements(F.true_pose_step(orc.x_gt, ut), lm) if a[2])
        ints = rs.randint(-1000, 1000, size=(64, 3, 12)); nrm = rs.randn(64, 2 * nv)
        t0 = time.perf_counter(); orc.step(ut, ints, nrm, rs.rand()); tt.append(time.perf_counter() - t0)
    out["cpu_baseline"] = {"value": 64 / tt[-1], "unit": "particle-steps/s", "cores": 1, "kind": "port",
                           "sample": "64 particles x %d landmarks, fourth step in %.2f s" % (NL, tt[-1])}
import json
print(json.dumps(out))
