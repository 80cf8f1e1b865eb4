"""The per-GPU share of the largest configuration -- a 16384^2 synthetic grid, 360 beams, 2^23 particles in float32
state -- on one GPU: the map structures at that size (1.0e9 neighbourhood entries behind 32-bit offsets) against
the oracle on a subset, the whole update through its size-independent properties, the cost of set_map, and the
refusal just past the 2^32-entry limit."""
import time

import numpy as np
import pytest

from oracle import mcl_oracle as O

pytestmark = pytest.mark.gpu

ALPHA = [4, 2, 4, 4, 4, 2]
N, B, M = 16384, 360, 1 << 23


def _local_lut(lut, cell, poses, reach):
    """The obstacle cells that can matter for `poses` (within `reach` of one of them in x and y), in list order: the
    oracle's answer for these poses is the same as with the whole list, whose 10^7 entries it would scan per pose."""
    keep = np.zeros(len(lut), bool)
    for p in poses:
        keep |= (np.abs(lut[:, 0] - p[0]) < reach) & (np.abs(lut[:, 1] - p[1]) < reach)
    return lut[keep], cell[keep]


def _free_device_bytes():
    """hipMemGetInfo of the HIP runtime the library itself is linked against (already loaded with it)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


@pytest.fixture(scope="module")
def huge():
    from slam_amd import _lib, workload as W
    grid = W.synthetic_map(N)
    occ = float(np.mean(grid == 0))
    assert 0.035 < occ < 0.05
    sp = O.SensorParams()
    ctx = _lib.Context(0, _lib.F32)
    free0 = _free_device_bytes()
    t0 = time.perf_counter()
    ctx.set_map(grid, W.RES, W.RES)
    ctx.set_sensor(W.beam_angles(B), sp.z_max, sp.sigma_hit, sp.lambda_short, [sp.w_hit, sp.w_short, sp.w_max, sp.w_rand],
                   5 * np.pi / 180.0)
    dt = time.perf_counter() - t0
    free1 = _free_device_bytes()
    return {"grid": grid, "ctx": ctx, "sp": sp, "set_map_s": dt, "map_bytes": free0 - free1, "occ": occ}


def test_map_structures_cost(huge):
    """Neighbourhood lists of the 16384^2 map at 4 % occupancy: ~0.9e9 entries of four bytes plus offsets and bit
    planes; built in well under a minute (host tile lists + device fill)."""
    print("set_map + set_sensor: %.1f s, %.2f GB of device memory, occupancy %.2f %%"
          % (huge["set_map_s"], huge["map_bytes"] / 1e9, 100 * huge["occ"]))
    assert huge["map_bytes"] < 7.0e9
    assert huge["set_map_s"] < 60.0


def test_sector_query_and_weights_against_the_oracle(huge):
    from slam_amd import _lib, workload as W
    ctx, grid, sp = huge["ctx"], huge["grid"], huge["sp"]
    angles = W.beam_angles(B)
    X = W.uniform_particles(grid, M)
    huge["X"] = X
    z_meas = ctx.raycast(X[0], want_hits=False)[0][0]
    huge["z_meas"] = z_meas
    ctx.set_particles(X)
    ctx.shard_sample(1.5, 0.0, 0.1, ALPHA, z_meas, None, 31, _lib.W_LOG)       # float32 state, device noise, log-weights
    Xd, _, lwd = ctx.get_particles(True, False, True)
    rng = np.random.RandomState(4)
    # 64 particles: some anywhere, some at the far end of the map (largest cell coordinates, largest list offsets)
    far = np.argsort(Xd[:, 0] + Xd[:, 1])[-200000:]
    pick = np.concatenate([rng.choice(M, 40, replace=False), rng.choice(far, 24, replace=False)])
    P = Xd[pick]
    lut, cell = O.obstacle_lut(grid)
    assert len(lut) > 1.0e7
    llut, lcell = _local_lut(lut, cell, P, 5.5)
    R, Hc = ctx.raycast(P)
    Ro, Ho = O.sector_query_batch(P, llut, lcell, angles)
    assert np.array_equal(R, Ro), int(np.sum(R != Ro))
    assert np.array_equal(Hc.astype(np.int64), Ho)
    assert Ho.max() > 2 ** 27          # hit cells are row * W + col of a 2^28-cell map
    pm = O.free_cell(P[:, 0], P[:, 1], grid)
    _, lwo, _ = O.weights_from_ranges(Ro, z_meas, pm, sp)
    both = np.isfinite(lwd[pick]) & np.isfinite(lwo)
    assert both.sum() >= 40
    err = np.abs(lwd[pick][both] - lwo[both]) / np.maximum(1.0, np.abs(lwo[both]))
    assert err.max() < 1e-5, err.max()
    # all particles: log-weights finite exactly where the moved pose stands on a free cell (up to float32 rounding
    # of the stored pose at a cell boundary)
    pm_all = O.free_cell(Xd[:, 0], Xd[:, 1], grid)
    assert np.mean(np.isfinite(lwd) != (pm_all == 1.0)) < 2e-3
    assert not np.any(np.isnan(lwd)) and not np.any(lwd == np.inf)


def test_whole_update_properties(huge):
    from slam_amd import _lib
    ctx, X, z_meas = huge["ctx"], huge["X"], huge["z_meas"]
    ctx.set_particles(X)
    ctx.sample(1.5, 0.0, 0.1, ALPHA, z_meas, None, seed=77)
    total = ctx.normalise(_lib.W_LOG)
    _, wn, _ = ctx.get_particles(False, True, False)
    assert abs(np.sum(wn) - 1.0) < 1e-9 and total > 0
    r = 0.61 / M
    idx = ctx.resample(r)
    assert idx.min() >= 0 and idx.max() <= M - 1 and np.all(np.diff(idx) >= 0)
    assert np.array_equal(idx, O.resample_indices(wn, r, clamp=True))          # the sequential walk's indices, 2^23 of them
    _, wr, _ = ctx.get_particles(False, True, False)
    assert np.array_equal(wr, wn[idx])
    ctx.close()


def test_map_beyond_the_list_limit_is_refused():
    """With every fourth cell an obstacle the neighbourhood lists of a 16384^2 map need more than 2^32 entries (a
    query tile of 8x8 cells sees about 5100 cells within reach, 4.2e6 tiles, a quarter of the cells occupied): a clear
    refusal, no partial state."""
    from slam_amd import _lib, workload as W
    grid = np.full((N, N), 255, np.uint8)
    grid[::2, ::2] = 0
    sp = O.SensorParams()
    ctx = _lib.Context(0, _lib.F32)
    ctx.set_sensor(W.beam_angles(B), sp.z_max, sp.sigma_hit, sp.lambda_short, [sp.w_hit, sp.w_short, sp.w_max, sp.w_rand],
                   5 * np.pi / 180.0)
    with pytest.raises(_lib.MclHipError, match="2\\^32"):
        ctx.set_map(grid, W.RES, W.RES)
    with pytest.raises(_lib.MclHipError):
        ctx.raycast(np.array([[10.0, 10.0, 0.0]]))
    ctx.close()
