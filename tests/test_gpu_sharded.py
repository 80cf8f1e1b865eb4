"""One filter over two ranks sharing the test box's single GPU (gloo carries the collectives):
the sharded update through the C ABI must equal the single-context update bit for bit."""
import os

import numpy as np
import pytest

from shard_helpers import run_ranks
from util import GOLDEN

pytestmark = pytest.mark.gpu

ALPHA = [4, 2, 4, 4, 4, 2]
ANGLES = [0, 15, 30, 45, 60, 75, 90, 345, 330, 315, 300, 285, 270]


def _setup(dtype):
    from slam_amd import _lib
    grid = np.load(os.path.join(GOLDEN, "grid_map_g.npy"))
    ctx = _lib.Context(0, dtype)
    ctx.set_map(grid, 0.15, 0.15)
    w4 = np.array([0.97, 0.005, 0.020, 0.005])
    ctx.set_sensor(ANGLES, 5, 0.15, 1.0, w4 / w4.sum(), 5 * np.pi / 180.0)
    return ctx


def _inputs(M):
    rng = np.random.RandomState(3)
    X = np.stack([rng.normal(3.0, 0.5, M), rng.normal(4.5, 0.5, M), rng.normal(0, 0.6, M)], 1)
    noise = rng.randint(-1000, 1001, size=(M, 3, 12)).astype(np.int16)
    return X, noise, rng.rand() / M


def _rank_fn(rank, world, dist, sizes, dtype, mode, z_meas, use_device_noise):
    from slam_amd.sharded import ShardedFilter
    M = sum(sizes)
    X, noise, r = _inputs(M)
    first = sum(sizes[:rank])
    sl = slice(first, first + sizes[rank])
    ctx = _setup(dtype)
    ctx.set_particles(X[sl])
    f = ShardedFilter(ctx, dist, rank, world, sizes[rank], weight_mode=mode)
    f.step(1.5, 0.0, 0.1, ALPHA, z_meas, r, seed=99, noise_ints=None if use_device_noise else noise[sl], want_idx=True)
    Xo, wo, _ = ctx.get_particles()
    return {"idx": f.last_idx, "X": Xo, "w": wo, "total": f.total}


@pytest.mark.parametrize("sizes,dtype,mode,devnoise", [((16384, 8192), 1, 1, False), ((8192, 8192), 1, 0, False),
                                                       ((8192, 11000), 0, 1, True)])
def test_two_ranks_equal_one_context(sizes, dtype, mode, devnoise):
    from slam_amd import _lib
    M = sum(sizes)
    X, noise, r = _inputs(M)
    ref = _setup(dtype)
    z_meas = ref.raycast(np.array([3.0, 4.5, 0.0]), want_hits=False)[0][0]
    ref.set_particles(X)
    ref.sample(1.5, 0.0, 0.1, ALPHA, z_meas, None if devnoise else noise, seed=99)
    total = ref.normalise(mode)
    idx = ref.resample(r, _lib.RESAMPLE_PARALLEL)
    Xr, wr, _ = ref.get_particles()
    res = run_ranks(_rank_fn, len(sizes), sizes, dtype, mode, z_meas, devnoise)
    assert all(x["total"] == total for x in res)
    assert np.array_equal(np.concatenate([x["idx"] for x in res]), idx)
    assert np.array_equal(np.concatenate([x["X"] for x in res]), Xr)
    assert np.array_equal(np.concatenate([x["w"] for x in res]), wr)


def _rccl_rank_fn(rank, world, dist, M, z_meas):
    from slam_amd import _lib
    from slam_amd.sharded import ShardedFilter
    X, noise, r = _inputs(M)
    ctx = _setup(_lib.F64)
    ctx.set_particles(X)
    f = ShardedFilter(ctx, dist, 0, 1, M, weight_mode=_lib.W_LOG, force_collectives=True)
    assert f.coll.backend == "nccl" and f.coll.dev.type == "cuda"
    f.step(1.5, 0.0, 0.1, ALPHA, z_meas, r, seed=99, noise_ints=noise, want_idx=True)
    Xo, wo, _ = ctx.get_particles()
    return {"idx": f.last_idx, "X": Xo, "w": wo, "total": f.total}


def test_rccl_path_with_device_tensors_on_one_rank():
    """The cross-rank path under the "nccl" backend (RCCL) with a group of one, in a fresh process
    that brings torch's HIP runtime up first (as bench.py does): the collectives run on GPU
    tensors, and packed rows travel between torch's device memory and the library's through raw
    device pointers -- the pieces the gloo tests cannot reach.  Result: bit-equal to the fused step."""
    from slam_amd import _lib
    M = 24576
    X, noise, r = _inputs(M)
    ref = _setup(_lib.F64)
    z_meas = ref.raycast(np.array([3.0, 4.5, 0.0]), want_hits=False)[0][0]
    ref.set_particles(X)
    ref.sample(1.5, 0.0, 0.1, ALPHA, z_meas, noise, seed=99)
    total = ref.normalise(_lib.W_LOG)
    idx = ref.resample(r, _lib.RESAMPLE_PARALLEL)
    Xr, wr, _ = ref.get_particles()
    res = run_ranks(_rccl_rank_fn, 1, M, z_meas, backend="nccl")[0]
    assert res["total"] == total
    assert np.array_equal(res["idx"], idx)
    assert np.array_equal(res["X"], Xr) and np.array_equal(res["w"], wr)
