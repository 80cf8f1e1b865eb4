"""One filter over several GPUs: one process per GPU, contiguous particle ranges per rank,
map and beam table replicated.  With a single rank this is the plain fused step."""
from . import _lib


class ShardedFilter:
    def __init__(self, ctx, dist, rank, world, m_local):
        self.ctx, self.dist, self.rank, self.world, self.m_local = ctx, dist, rank, world, m_local
        if world > 1:
            raise NotImplementedError("multi-GPU exchange not built yet")

    def step(self, v, omega, dt, alpha, z_meas, r, seed):
        self.ctx.step(v, omega, dt, alpha, z_meas, r, None, seed, _lib.W_LOG, wait=False)
