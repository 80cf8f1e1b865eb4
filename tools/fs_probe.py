import os, sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import fastslam_oracle as F
from slam_amd import _lib as L
from test_gpu_fastslam import _truth
M, nl, cap = 1 << 20, 256, 320
rng = np.random.RandomState(9)
lm = np.stack([rng.uniform(-8, 33, nl), rng.uniform(-24, 8, nl), np.zeros(nl)], 1).astype(np.float32)
ctx = L.Context(0, L.F64)
ctx.fs_init(M, cap)
x_gt = [0, 0, 0]
for t in range(6):
    ut = [25, 0.001] if t % 5 else [25, -np.pi / 2 * 5]
    x_gt = F.true_pose_step(x_gt, ut)
    tr, tp, tv = _truth(x_gt, lm)
    idx = ctx.fs_step(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, None, None, 0.37 / M, seed=100 + t, want_idx=True)
    nf, _ = ctx.fs_feature_counts()
    print("step", t, "sensed", int(tv.sum()), "unique ancestors", len(np.unique(idx)), "idx range", idx.min(), idx.max(), "mean features %.1f" % nf.mean())
