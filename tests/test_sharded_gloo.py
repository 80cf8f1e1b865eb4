"""The cross-rank bookkeeping of one filter over several ranks (slam_amd/sharded.py) with
world_size 2 and 3 on CPU (gloo): global sum in numpy's order, global record list, boundary
counts, slice plan and exchange must reproduce the single-process result bit for bit."""
import numpy as np
import pytest

from oracle import mcl_oracle as O
from shard_helpers import NumpyShardBackend, run_ranks, synthetic_logw

ALPHA = [4, 2, 4, 4, 4, 2]


def _inputs(M, kind, seed=0):
    rng = np.random.RandomState(seed)
    X = np.stack([rng.normal(3.0, 0.3, M), rng.normal(4.0, 0.3, M), rng.uniform(-3, 3, M)], 1)
    if kind == "peaky":          # one dominant particle in the middle: most positions draw from one source
        X[M // 2] = [3.0, 4.0, np.pi / 2]
        X[:, 0] += 0.6
        X[M // 2, 0] -= 0.6
    noise = rng.randint(-1000, 1001, size=(M, 3, 12)).astype(np.int16)
    if kind == "peaky":
        noise[M // 2] = 0
    r = rng.rand() / M
    return X, noise, r


def _rank_fn(rank, world, dist, sizes, kind, mode):
    from slam_amd.sharded import ShardedFilter
    M = sum(sizes)
    X, noise, r = _inputs(M, kind)
    first = sum(sizes[:rank])
    sl = slice(first, first + sizes[rank])
    b = NumpyShardBackend(X[sl])
    f = ShardedFilter(b, dist, rank, world, sizes[rank], weight_mode=mode)
    f.step(1.5, 0.0, 0.1, ALPHA, None, r, seed=0, noise_ints=noise[sl], want_idx=True)
    return {"idx": f.last_idx, "X": b.X, "w": b.w, "total": f.total}


@pytest.mark.parametrize("sizes,kind,mode", [
    ((8192, 8192), "spread", 1),
    ((8192, 8192), "peaky", 1),
    ((8192, 5000), "spread", 0),
    ((8192, 16384, 3000), "spread", 1),
])
def test_sharded_step_matches_single_process(sizes, kind, mode):
    world = len(sizes)
    M = sum(sizes)
    res = run_ranks(_rank_fn, world, sizes, kind, mode)
    # single-process result
    X, noise, r = _inputs(M, kind)
    grid = np.full((200, 200), 255, np.uint8)
    Xm, pm = O.motion(X, (1.5, 0.0), 0.1, grid, noise, tuple(ALPHA))
    lw = synthetic_logw(Xm)
    w = np.exp(lw - np.max(lw)) if mode == 1 else np.exp(lw) * pm
    total = np.sum(w)
    wn = w / total
    idx = O.resample_indices(wn, r, clamp=True)
    got_idx = np.concatenate([x["idx"] for x in res])
    assert all(x["total"] == total for x in res), "global sum differs from np.sum of the whole vector"
    assert np.array_equal(got_idx, idx)
    assert np.array_equal(np.concatenate([x["X"] for x in res]), Xm[idx])
    assert np.array_equal(np.concatenate([x["w"] for x in res]), wn[idx])


def _misaligned_fn(rank, world, dist):
    from slam_amd.sharded import ShardedFilter
    try:
        ShardedFilter(NumpyShardBackend(np.zeros((1000, 3))), dist, rank, world, 1000)
    except ValueError:
        return "raised"
    return "no error"


def test_uneven_shards_must_align_with_sum_chunks():
    assert run_ranks(_misaligned_fn, 2) == ["raised", "raised"]
