"""The per-GPU share of the largest configuration -- a 16384^2 synthetic grid, 360 beams, 2^23 particles in float32
state -- on one GPU: the map structures at that size (1.0e9 neighbourhood entries) against the oracle on a subset,
the whole update through its size-independent properties, the cost of set_map; the same map size with a fifth of
its cells occupied (4.6e9 entries: past what 32-bit offsets address), and the refusal of lists that do not fit the
device memory allowed for them."""
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
    """hipMemGetInfo of a HIP runtime this process has loaded and initialised: the one the library is linked against
    (under /opt/rocm) first -- a process that has imported torch holds torch's own copy too, and a bare
    dlopen("libamdhip64.so") may then come back with an instance that sees no device."""
    import ctypes
    paths = {line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line}
    assert paths, "no HIP runtime loaded"
    codes = []
    for path in sorted(paths, key=lambda q: (not q.startswith("/opt/rocm"), q)):
        hip = ctypes.CDLL(path)
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        rc = hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
        if rc == 0:
            return free.value
        codes.append((path, rc))
    raise AssertionError("hipMemGetInfo failed: %r" % (codes,))


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
    """Neighbourhood lists of the 16384^2 map at 4 % occupancy, one per 4 x 4-cell query tile: ~3.3e9 entries of four
    bytes plus offsets, tile records and bit planes; built in well under a minute (host tile lists + device fill)."""
    print("set_map + set_sensor: %.1f s, %.2f GB of device memory, occupancy %.2f %%"
          % (huge["set_map_s"], huge["map_bytes"] / 1e9, 100 * huge["occ"]))
    assert huge["map_bytes"] < 20.0e9
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


def test_dense_map_past_four_billion_list_entries():
    """A 16384^2 map with a fifth of its cells occupied: the neighbourhood lists hold more than 2^32 entries (64-bit
    offsets; the reference's make_obstacle_lut, mc_localization.py:61-71, has no limit of its own).  Stated bounds:
    under 80 GB of device memory and under three minutes for set_map + set_sensor; ranges, hit cells and log-weights
    of poses at the far end of the map -- the largest list offsets -- against the oracle."""
    from slam_amd import _lib, workload as W
    grid = W.synthetic_map(N, target_occupancy=0.2)
    occ = float(np.mean(grid == 0))
    assert 0.2 <= occ < 0.22
    sp = O.SensorParams()
    Bd = 36
    angles = W.beam_angles(Bd)
    ctx = _lib.Context(0, _lib.F32)
    free0 = _free_device_bytes()
    t0 = time.perf_counter()
    ctx.set_map(grid, W.RES, W.RES)
    ctx.set_sensor(angles, sp.z_max, sp.sigma_hit, sp.lambda_short, [sp.w_hit, sp.w_short, sp.w_max, sp.w_rand], 5 * np.pi / 180.0)
    dt = time.perf_counter() - t0
    used = free0 - _free_device_bytes()
    print("16384^2 at %.1f %% occupancy: set_map + set_sensor %.1f s, %.2f GB of device memory" % (100 * occ, dt, used / 1e9))
    assert used > 4.0 * (1 << 32) and used < 80e9          # more than 2^32 entries of four bytes
    assert dt < 180.0
    # poses in the last rows and columns of the map (offsets beyond 2^32 entries) and a few near its origin
    rng = np.random.RandomState(8)
    ext = N * W.RES
    P = np.concatenate([np.stack([rng.uniform(ext - 40.0, ext - 1.0, 24), rng.uniform(ext - 40.0, ext - 1.0, 24), rng.uniform(0, 6.2, 24)], 1),
                        np.stack([rng.uniform(1.0, 30.0, 8), rng.uniform(1.0, 30.0, 8), rng.uniform(0, 6.2, 8)], 1)])
    P = P.astype(np.float32).astype(np.float64)
    llut, lcell = [], []
    for (r0, c0) in ((N - 320, N - 320), (0, 0)):           # the obstacle cells of the two corners, in list order
        sub = grid[r0:r0 + 320, c0:c0 + 320]
        lu, ce = O.obstacle_lut(sub)
        ii, jj = ce // 320 + r0, ce % 320 + c0
        llut.append(np.stack([(jj.astype(np.float32) * np.float32(W.RES)), (ii.astype(np.float32) * np.float32(W.RES))], 1))   # float32, as obstacle_lut forms them
        lcell.append(ii.astype(np.int64) * N + jj)
    order = np.argsort(np.concatenate(lcell), kind="stable")
    llut, lcell = np.concatenate(llut)[order], np.concatenate(lcell)[order]
    R, Hc = ctx.raycast(P)
    Ro, Ho = O.sector_query_batch(P, llut, lcell, angles)
    assert np.array_equal(R, Ro), int(np.sum(R != Ro))
    assert np.array_equal(Hc.astype(np.int64), Ho)
    # the float32 throughput path (interval-only kernel, records with 64-bit list offsets) on the same poses
    z_meas = R[0]
    M2 = 8192
    X = np.tile(P, (M2 // len(P), 1))
    ctx.set_particles(X)
    ctx.shard_sample(0.0, 1e-12, 0.1, ALPHA, z_meas, np.zeros((M2, 3, 12), np.int16), 0, _lib.W_LOG)
    lwd = ctx.get_particles(False, False, True)[2][:len(P)]
    _, lwo, _ = O.weights_from_ranges(Ro, z_meas, O.free_cell(P[:, 0], P[:, 1], grid), sp)
    both = np.isfinite(lwd) & np.isfinite(lwo)
    assert np.array_equal(np.isfinite(lwd), np.isfinite(lwo)) and both.sum() >= 8
    err = np.abs(lwd[both] - lwo[both]) / np.maximum(1.0, np.abs(lwo[both]))
    assert err.max() < 1e-5, err.max()
    ctx.close()


def test_lists_beyond_the_memory_allowed_are_refused(monkeypatch):
    """The neighbourhood lists are refused, before anything is allocated, when they would not fit the device memory
    that is free (or the budget MCLHIP_LIST_BUDGET_GB sets): a clear message, no partial state."""
    from slam_amd import _lib, workload as W
    monkeypatch.setenv("MCLHIP_LIST_BUDGET_GB", "0.05")
    grid = W.synthetic_map(2048)
    sp = O.SensorParams()
    ctx = _lib.Context(0, _lib.F32)
    ctx.set_sensor(W.beam_angles(B), sp.z_max, sp.sigma_hit, sp.lambda_short, [sp.w_hit, sp.w_short, sp.w_max, sp.w_rand],
                   5 * np.pi / 180.0)
    free0 = _free_device_bytes()
    with pytest.raises(_lib.MclHipError, match="neighbourhood lists of this map need"):
        ctx.set_map(grid, W.RES, W.RES)
    assert free0 - _free_device_bytes() < 64 << 20
    with pytest.raises(_lib.MclHipError):
        ctx.raycast(np.array([[10.0, 10.0, 0.0]]))
    monkeypatch.setenv("MCLHIP_LIST_BUDGET_GB", "2")
    ctx.set_map(grid, W.RES, W.RES)                       # with room for them the same map is accepted
    assert np.all(ctx.raycast(np.array([[10.0, 10.0, 0.0]]))[0] <= W.Z_MAX)
    ctx.close()
