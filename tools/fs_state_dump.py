"""Runs FastSLAM with device noise on the dense landmark field and saves the particle state (for comparing two
builds of the library: MCLHIP_LIBNAME selects the build)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import fast_slam1 as S
M = int(os.environ.get("FS_M", 1 << 16)); NL = 256; steps = int(os.environ.get("FS_STEPS", 6))
np.random.seed(123)
rng = np.random.RandomState(9)
lm = np.stack([rng.uniform(-8, 33, NL), rng.uniform(-24, 8, NL), np.zeros(NL)], 1).astype(np.float32)
slam = S.FastSLAM(M=M, landmarks=lm, capacity=NL + 64, noise="device")
for t in range(steps):
    slam.control([25, 0.001] if t % 5 else [25, -np.pi / 2 * 5])
st = slam.state()
np.savez(sys.argv[1], **{k: st[k] for k in ("pose", "w", "nfeat", "new_id", "f_m", "f_cov", "f_i", "f_w", "f_id")})
print("mean features", st["nfeat"].mean())
