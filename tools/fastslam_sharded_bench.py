"""FastSLAM 1.0 at the large configuration through the multi-GPU code path on ONE rank (RCCL group of one): what the
exchanges cost per step, device-driven against host-combined.  FS_PATH=device|host."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from slam_amd import _lib as L
from slam_amd.sharded import ShardedFastSLAM
from oracle import fastslam_oracle as F          # the true pose and its noise-free readings (host side of the step)
M = int(os.environ.get("FS_M", 1 << 20)); NL = int(os.environ.get("FS_L", 256)); steps = int(os.environ.get("FS_STEPS", 8))
path = os.environ.get("FS_PATH", "device")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
rng = np.random.RandomState(9)
lm = np.stack([rng.uniform(-8, 33, NL), rng.uniform(-24, 8, NL), np.zeros(NL)], 1).astype(np.float32)
ctx = L.Context(0, L.F64)
f = ShardedFastSLAM(ctx, dist, 0, 1, M, NL + 64, device_path=(path == "device"))
x_gt = [0, 0, 0]
rs = np.random.RandomState(123)
t_all = []
for t in range(steps):
    ut = [25, 0.001] if t % 5 else [25, -np.pi / 2 * 5]
    x_gt = F.true_pose_step(x_gt, ut)
    dx = lm[:, 0].astype(np.float64) - x_gt[0]; dy = lm[:, 1].astype(np.float64) - x_gt[1]
    tr = np.sqrt(dx * dx + dy * dy); tp = np.arctan2(dy, dx) - x_gt[2]
    tv = ((0 < tr) & (tr <= 10.0)).astype(np.uint8)
    ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    f.step(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, None, None, rs.rand() / M, seed=t + 1)
    ctx.sync(); torch.cuda.synchronize()
    t_all.append(time.perf_counter() - t0)
    print("step", t, "%.2f ms" % (t_all[-1] * 1e3), flush=True)
ms = float(np.median(t_all[2:])) * 1e3
print(json.dumps({"metric": "FastSLAM 1.0 multi-GPU code path on one rank", "path": path, "particles": M, "landmarks": NL,
                  "ms_per_step": ms, "rows_per_step": f.rows_sent_total / max(f.steps_done, 1) if hasattr(f, "steps_done") else None}))
dist.destroy_process_group()
