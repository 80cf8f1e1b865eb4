"""The benchmark configuration at full size (2^20 particles x 180 beams, 4096^2 synthetic grid),
checked through properties that do not need the oracle to run a million particles: a random
subset against the oracle, order-independence, systematic-resampling counts, monotone indices."""
import numpy as np
import pytest

from oracle import mcl_oracle as O

pytestmark = pytest.mark.gpu

ALPHA = [4, 2, 4, 4, 4, 2]


@pytest.fixture(scope="module")
def big():
    from slam_amd import _lib, workload as W
    grid = W.synthetic_map(4096)
    angles = W.beam_angles(180)
    M = 1 << 20
    X = W.uniform_particles(grid, M)
    sp = O.SensorParams()

    def make(dtype):
        ctx = _lib.Context(0, dtype)
        ctx.set_map(grid, W.RES, W.RES)
        ctx.set_sensor(angles, sp.z_max, sp.sigma_hit, sp.lambda_short, [sp.w_hit, sp.w_short, sp.w_max, sp.w_rand],
                       5 * np.pi / 180.0)
        return ctx
    return {"grid": grid, "angles": angles, "X": X, "make": make, "M": M, "sp": sp}


def test_subset_of_a_million_particles_matches_oracle(big):
    from slam_amd import _lib
    ctx = big["make"](_lib.F64)
    X, M = big["X"], big["M"]
    z_meas = ctx.raycast(X[0], want_hits=False)[0][0]
    rng = np.random.RandomState(11)
    noise = rng.randint(-1000, 1001, size=(M, 3, 12)).astype(np.int16)
    ctx.set_particles(X)
    ctx.sample(1.5, 0.5, 0.1, ALPHA, z_meas, noise)
    Xd, wd, lwd = ctx.get_particles(True, True, True)
    pick = rng.choice(M, 48, replace=False)
    lut, cell = O.obstacle_lut(big["grid"])
    Xo, pmo = O.motion(X[pick], (1.5, 0.5), 0.1, big["grid"], noise[pick])
    assert np.max(np.abs(Xd[pick] - Xo) / np.maximum(1, np.abs(Xo))) < 1e-12
    R, Hc = ctx.raycast(Xd[pick])
    Ro, Ho = O.sector_query_batch(Xd[pick], lut, cell, big["angles"])
    assert np.array_equal(R, Ro) and np.array_equal(Hc.astype(np.int64), Ho)
    _, lwo, _ = O.weights_from_ranges(Ro, z_meas, O.free_cell(Xd[pick, 0], Xd[pick, 1], big["grid"]), big["sp"])
    fin = np.isfinite(lwo)
    assert np.array_equal(np.isfinite(lwd[pick]), fin)
    assert np.allclose(lwd[pick][fin], lwo[fin], rtol=1e-12, atol=1e-10)


def test_results_do_not_depend_on_particle_order(big):
    from slam_amd import _lib
    ctx = big["make"](_lib.F32)
    X, M = big["X"], big["M"]
    z_meas = ctx.raycast(X[0], want_hits=False)[0][0]
    n = 1 << 17
    rng = np.random.RandomState(5)
    noise = rng.randint(-1000, 1001, size=(n, 3, 12)).astype(np.int16)
    perm = rng.permutation(n)
    ctx.set_particles(X[:n])
    ctx.sample(1.5, 0.0, 0.1, ALPHA, z_meas, noise)
    a = ctx.get_particles(True, False, True)
    ctx.set_particles(X[:n][perm])
    ctx.sample(1.5, 0.0, 0.1, ALPHA, z_meas, noise[perm])
    b = ctx.get_particles(True, False, True)
    assert np.array_equal(a[0][perm], b[0]) and np.array_equal(a[2][perm], b[2])


def test_full_step_resampling_properties(big):
    from slam_amd import _lib
    ctx = big["make"](_lib.F32)
    X, M = big["X"], big["M"]
    z_meas = ctx.raycast(X[0], want_hits=False)[0][0]
    ctx.set_particles(X)
    ctx.sample(1.5, 0.0, 0.1, ALPHA, z_meas, None, seed=77)
    total = ctx.normalise(_lib.W_LOG)
    _, wn, _ = ctx.get_particles(False, True, False)
    assert total == np.sum(wn * total) or abs(np.sum(wn) - 1.0) < 1e-9
    r = 0.37 / M
    idx = ctx.resample(r)
    assert idx.min() >= 0 and idx.max() <= M - 1
    assert np.all(np.diff(idx) >= 0)
    # systematic resampling: particle i is drawn floor(M w_i) or ceil(M w_i) times (+-1 for rounding)
    counts = np.bincount(idx, minlength=M)
    expect = wn * M
    assert np.all(np.abs(counts - expect) < 1.0 + 1e-6 * expect + 1e-6)
    # and exactly the indices of the sequential walk
    assert np.array_equal(idx, O.resample_indices(wn, r, clamp=True))
    Xr, wr, _ = ctx.get_particles()
    assert np.array_equal(wr, wn[idx])
