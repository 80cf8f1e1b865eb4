"""Determinism probe for FastSLAM with device noise: the same two steps run twice in fresh contexts; the state
after the per-particle kernels of step 2 (no resampling) and after the whole step 2 are compared."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib as L
from oracle import fastslam_oracle as F
M = int(os.environ.get("FS_M", 1 << 16)); NL = 256
rng = np.random.RandomState(9)
lm = np.stack([rng.uniform(-8, 33, NL), rng.uniform(-24, 8, NL), np.zeros(NL)], 1).astype(np.float32)
KEYS = ("pose", "w", "nfeat", "new_id", "f_m", "f_cov", "f_i", "f_w", "f_id")


def truth(x_gt):
    t = F.true_measurements(list(x_gt), lm)
    return np.array([float(a[0]) for a in t]), np.array([float(a[1]) for a in t]), np.array([a[2] for a in t], np.uint8)


def run(kernel_only_last):
    ctx = L.Context(0, L.F64)
    ctx.fs_init(M, NL + 64)
    x_gt = [0, 0, 0]
    out = []
    for t in range(2):
        ut = [25, 0.001] if t % 5 else [25, -np.pi / 2 * 5]
        x_gt = F.true_pose_step(x_gt, ut)
        tr, tp, tv = truth(x_gt)
        if t == 1 and kernel_only_last:
            ctx.fs_shard_kernel(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, None, None, seed=t + 1)
            idx = None
        else:
            idx = ctx.fs_step(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, None, None, 0.37 / M, seed=t + 1, want_idx=True)
        st = ctx.fs_get_state()
        out.append((idx, {k: st[k].copy() for k in KEYS}))
    ctx.close()
    return out


for mode in (True, False):
    a, b = run(mode), run(mode)
    for t in range(2):
        diffs = [k for k in KEYS if not np.array_equal(a[t][1][k], b[t][1][k], equal_nan=True)]
        same_idx = a[t][0] is None or np.array_equal(a[t][0], b[t][0])
        print("kernel-only last step" if mode else "whole steps", "step", t, "differing fields:", diffs, "ancestors equal:", same_idx)
        if diffs:
            bad = np.nonzero(a[t][1]["w"] != b[t][1]["w"])[0]
            print("   particles with different weight:", len(bad), bad[:8])
