"""Parity of the histogram filter on the GPU (through the C ABI) with the golden vectors captured
from the reference's Robot and with the oracle.  Needs an MI355X: `-m gpu`.

Bars.  Initial belief and update (from the reference's own pk_pred): bit-exact float32, the float32
sum included.  predict: the zero pattern is exact; values within 1e-8 relative element by element.  The motion density divides by a guarded 1e-6 and subtracts two
arctangents of nearly equal angles (robot.py:97-109): a one-ulp difference between numpy's and the
device's atan2 is amplified to ~3e-10 relative in the affected terms, so no independent
implementation can hold 1e-12 there (measured here: 2.5e-10 at worst on the reference's map).  Chained predict+update:
within one float32 ulp.
"""
import numpy as np
import pytest

from oracle import grid_oracle as G
from oracle import mcl_oracle as O
from util import load, steps

pytestmark = pytest.mark.gpu

ALPHA = list(G.ALPHA)
HALF = 5 * np.pi / 180.0


def _ctx(grid, n_dir, res_theta, angles=G.SENSOR_ANGLES):
    from slam_amd import _lib as L
    ctx = L.Context(0, L.F64)
    ctx.grid_set_map(grid, G.RES_X, G.RES_Y, G.PIXEL_OBSTACLE)
    sp = O.SensorParams()
    ctx.grid_set_sensor(list(angles), n_dir, res_theta, sp.z_max, sp.sigma_hit, sp.lambda_short,
                        [sp.w_hit, sp.w_short, sp.w_max, sp.w_rand], HALF)
    return ctx


def _close_prediction(pp, ref):
    nz = ref != 0
    assert bool(np.array_equal(pp != 0, nz))
    if nz.any():
        worst = float(np.max(np.abs(pp - ref)[nz] / np.abs(ref[nz])))
        assert worst < 1e-8, worst


@pytest.mark.parametrize("name", ["grid_seed0", "grid_seed1"])
def test_reference_configuration(name):
    g = load(name)
    ctx = _ctx(g["map"], 4, 90, [int(a) for a in g["sensor_angles"]])
    ctx.grid_init_belief()
    assert np.array_equal(ctx.grid_get_belief()[0], g["pk0"])
    for s in steps(g):
        ctx.grid_set_belief(s["pk_in"])
        ctx.grid_predict(s["ut"][0], s["ut"][1], G.DT, ALPHA)
        pp = ctx.grid_get_belief(False, True)[1]
        _close_prediction(pp, s["pk_pred"])
        # update from the reference's own prediction: bit-exact float32
        ctx.grid_set_prediction(s["pk_pred"])
        total = ctx.grid_update(s["z_meas"])
        pk = ctx.grid_get_belief()[0]
        assert np.array_equal(pk, s["pk_out"]), "%d cells differ" % np.sum(pk != s["pk_out"])
        raw = G.update(s["pk_pred"], g["map"], [float(v) for v in s["z_meas"]], 4)[1]
        assert total == raw
        # update from the device's own prediction: within one float32 ulp
        ctx.grid_set_prediction(pp)
        ctx.grid_update(s["z_meas"])
        pk2 = ctx.grid_get_belief()[0].astype(np.float64)
        ref = s["pk_out"].astype(np.float64)
        assert np.all(np.abs(pk2 - ref) <= 1.3e-7 * np.abs(ref))


def test_dropin_class_follows_reference_run():
    """slam_amd.robot.Robot driven like grid_localization.py:25-34 under the same np.random seed."""
    from slam_amd.robot import Robot, res_x, res_y
    g = load("grid_seed0")
    np.random.seed(int(g["seed"]))
    x_pose = np.array([[20 * res_x, 30 * res_y, 0]], dtype=np.float32).T
    robot = Robot(g["map"], x_pose)
    assert np.array_equal(robot.pk, g["pk0"])
    for s in steps(g):
        robot.predict([1.5, 0])
        assert np.array_equal(robot.x_gt, s["x_gt_out"])
        _close_prediction(robot.pk_pred, s["pk_pred"])
        robot.update()
        assert np.array_equal(np.array([float(z) for z in robot.sensor_ranges]), s["z_meas"])
        ref = s["pk_out"].astype(np.float64)
        assert robot.pk.dtype == np.float32
        assert np.all(np.abs(robot.pk.astype(np.float64) - ref) <= 3e-7 * np.abs(ref))


def _small_map(n, seed):
    rng = np.random.RandomState(seed)
    m = np.full((n, n), 255, np.uint8)
    m[0, :] = m[-1, :] = 0
    m[:, 0] = m[:, -1] = 0
    for _ in range(n // 3):
        i, j = rng.randint(1, n - 1, 2)
        if rng.randint(2):
            m[i, j:j + rng.randint(2, 9)] = rng.choice([0, 127, 199])
        else:
            m[i:i + rng.randint(2, 9), j] = rng.choice([0, 127, 199])
    m[rng.randint(1, n - 1), rng.randint(1, n - 1)] = 200   # grey but free
    return m


@pytest.mark.parametrize("n,n_dir,ut", [(28, 8, (1.5, 0.0)), (22, 72, (1.5, 0.5)), (20, 6, (0.7, -0.3))])
def test_more_headings_against_the_oracle(n, n_dir, ut):
    """Eight, 72 (5 degree bins: 936 table entries, the large-LDS path) and six headings (the
    scalar stencil variant).  The reference itself is limited to four; the oracle's generalisation
    sums over all headings.

    With 5 degree bins and the reference's beam set (multiples of 15 degrees, +-5 degree sectors)
    sector edges fall on multiples of 5 degrees, among them the grid's own axis and diagonal
    directions: an obstacle cell exactly on such a line is in or out by the last bit of atan2, which
    numpy builds, libm versions and the device do not share.  The 72-heading case therefore shifts
    the beams by one degree, which removes the exact ties and leaves everything else as it is."""
    grid = _small_map(n, n_dir)
    res_theta = 360 // n_dir
    angles = [a + 1 for a in G.SENSOR_ANGLES] if n_dir == 72 else list(G.SENSOR_ANGLES)
    ctx = _ctx(grid, n_dir, res_theta, angles)
    rng = np.random.RandomState(n)
    pk = (rng.rand(n, n, n_dir) ** 3).astype(np.float32)
    pk[grid < 200] = 0
    pk /= pk.sum()
    ctx.grid_set_belief(pk)
    ctx.grid_predict(ut[0], ut[1], G.DT, ALPHA)
    pp = ctx.grid_get_belief(False, True)[1]
    ref = G.predict(pk, grid, ut, n_dir, res_theta=res_theta, neighbour_dirs=range(n_dir))
    _close_prediction(pp, ref)
    z_meas = [0.61, 1.2, 5.0, 2.35, 0.0, 4.99, 3.3, 0.9, 5.0, 1.75, 2.5, 0.3, 4.2]
    ii, jj, Z = G.predicted_ranges(grid, n_dir, angles=angles, res_theta=res_theta)
    want, total, _ = G.update(ref, grid, z_meas, n_dir, angles=angles, Z=Z)
    ctx.grid_set_prediction(ref)
    got_total = ctx.grid_update(z_meas)
    got = ctx.grid_get_belief()[0]
    assert got_total == total
    assert np.array_equal(got, want), "%d cells differ" % np.sum(got != want)


def _stage_ranges(grid, ii, jj, n_dir, res_theta, angles):
    """Ranges and hit cells of the device's stage sector query (mclhip_raycast) for the poses the histogram filter
    predicts from -- cell centres (j*res_x, i*res_y) at every heading -- and the oracle's for the same poses."""
    from slam_amd import _lib as L
    sp = O.SensorParams()
    m = L.Context(0, L.F64)
    m.set_map(grid, G.RES_X, G.RES_Y)                # obstacle == 0: the same cells as < 200 on a {0, 255} map
    m.set_sensor(list(angles), sp.z_max, sp.sigma_hit, sp.lambda_short, [sp.w_hit, sp.w_short, sp.w_max, sp.w_rand], HALF)
    N = len(ii)
    poses = np.zeros((N, n_dir, 3))
    poses[:, :, 0] = (jj * G.RES_X)[:, None]
    poses[:, :, 1] = (ii * G.RES_Y)[:, None]
    poses[:, :, 2] = np.array([G.heading(d, res_theta) for d in range(n_dir)])[None, :]
    P = poses.reshape(-1, 3)
    R_g, H_g = m.raycast(P)
    m.close()
    return P, R_g.reshape(N, n_dir, -1), H_g.astype(np.int64).reshape(N, n_dir, -1)


def _assert_only_lattice_ties(grid, ii, jj, res_theta, angles, R_g, H_g, R_o, H_o):
    """Every (cell, heading, beam) whose range differs between device and oracle is decided by ONE obstacle cell that
    lies exactly on a sector edge: on a row, column or diagonal through the pose cell (integer test) at exactly
    +-5 degrees from the beam axis.  There the bearing equals the edge to the last bit and atan2's rounding -- which
    numpy's SIMD path, libm and the device do not share -- decides; everywhere else the two agree.  Returns the count."""
    W = grid.shape[1]
    bad = np.argwhere(R_g != R_o)
    for n, d, k in bad:
        cid = H_g[n, d, k] if R_g[n, d, k] < R_o[n, d, k] else H_o[n, d, k]     # the cell one side let in and the other did not
        assert cid >= 0
        di, dj = cid // W - ii[n], cid % W - jj[n]
        assert di == 0 or dj == 0 or abs(di) == abs(dj), (n, d, k, di, dj)
        bearing = np.degrees(np.arctan2(float(np.sign(di)), float(np.sign(dj)))) % 360.0    # a multiple of 45 degrees, exact
        axis = (angles[k] + d * res_theta) % 360.0
        off = (axis - bearing) % 360.0
        assert off in (5.0, 355.0), (n, d, k, axis, bearing)
    return len(bad)


def test_72_headings_unshifted_differ_from_the_oracle_only_at_exact_ties():
    """The reference's own beam set with 5-degree heading bins, not shifted: sector edges then fall on the grid's
    axis and diagonal directions and cells exactly on an edge are in or out by the last bit of atan2.  Every
    difference between device and oracle is shown to be such a tie (and counted); with the tie decisions taken as the
    device takes them the update is bit-equal to the oracle's."""
    n, n_dir, res_theta = 22, 72, 5
    rng = np.random.RandomState(72)
    grid = np.full((n, n), 255, np.uint8)
    grid[0, :] = grid[-1, :] = 0
    grid[:, 0] = grid[:, -1] = 0
    for _ in range(n // 3):
        i, j = rng.randint(1, n - 1, 2)
        if rng.randint(2):
            grid[i, j:j + rng.randint(2, 9)] = 0
        else:
            grid[i:i + rng.randint(2, 9), j] = 0
    angles = list(G.SENSOR_ANGLES)
    ii, jj, Z_o = G.predicted_ranges(grid, n_dir, angles=angles, res_theta=res_theta)
    P, R_g, H_g = _stage_ranges(grid, ii, jj, n_dir, res_theta, angles)
    lut, cell = O.obstacle_lut(grid)
    R_o, H_o = O.sector_query_batch(P, lut, cell, angles)
    R_o, H_o = R_o.reshape(Z_o.shape), H_o.reshape(Z_o.shape)
    assert np.array_equal(R_o, Z_o)                               # the two oracles agree
    n_ties = _assert_only_lattice_ties(grid, ii, jj, res_theta, angles, R_g, H_g, R_o, H_o)
    print("72 headings unshifted, %dx%d map: %d of %d (cell, heading, beam) ranges differ, all exact lattice ties"
          % (n, n, n_ties, R_o.size))
    assert 0 < n_ties < 0.02 * R_o.size
    # the histogram filter's own kernel decides the ties as the stage query does: bit-equal update given those ranges
    ctx = _ctx(grid, n_dir, res_theta, angles)
    pred = rng.rand(n, n, n_dir) ** 2
    pred[grid < 200] = 0
    z_meas = [0.61, 1.2, 5.0, 2.35, 0.0, 4.99, 3.3, 0.9, 5.0, 1.75, 2.5, 0.3, 4.2]
    ctx.grid_set_prediction(pred)
    got_total = ctx.grid_update(z_meas)
    got = ctx.grid_get_belief()[0]
    want, total, _ = G.update(pred, grid, z_meas, n_dir, angles=angles, Z=R_g)
    assert got_total == total
    assert np.array_equal(got, want), "%d cells differ" % np.sum(got != want)
    # and against the oracle's own ranges the belief differs only where a tie sits
    want_o = G.update(pred, grid, z_meas, n_dir, angles=angles, Z=Z_o)[2]
    raw_g = G.update(pred, grid, z_meas, n_dir, angles=angles, Z=R_g)[2]
    tied = np.zeros((n, n, n_dir), bool)
    tied[ii[:, None], jj[:, None], np.arange(n_dir)[None, :]] = np.any(R_g != Z_o, axis=2)
    assert not np.any((raw_g != want_o) & ~tied)


def test_entry_classes_with_negative_and_irregular_angles():
    """Beams below zero (their table angle reaches the reference's test one turn low), beams that no other
    heading shares and beams that coincide across headings only after a full turn: the update on classes of
    entries equals the oracle's entry-by-entry form.  Angles whose sector would leave (-2pi, 4pi) are refused."""
    n, n_dir, res_theta = 24, 8, 45
    grid = _small_map(n, n_dir)
    angles = [-91.0, -44.0, -3.5, 1.0, 46.0, 91.0, 133.7, 181.0, 226.0, 271.0, 316.0, 359.0, 12.25]
    ctx = _ctx(grid, n_dir, res_theta, angles)
    rng = np.random.RandomState(5)
    pred = rng.rand(n, n, n_dir) ** 2
    pred[grid < 200] = 0
    z_meas = [0.61, 1.2, 5.0, 2.35, 0.0, 4.99, 3.3, 0.9, 5.0, 1.75, 2.5, 0.3, 4.2]
    ii, jj, Z = G.predicted_ranges(grid, n_dir, angles=angles, res_theta=res_theta)
    want, total, _ = G.update(pred, grid, z_meas, n_dir, angles=angles, Z=Z)
    ctx.grid_set_prediction(pred)
    got_total = ctx.grid_update(z_meas)
    got = ctx.grid_get_belief()[0]
    assert got_total == total
    assert np.array_equal(got, want), "%d cells differ" % np.sum(got != want)
    from slam_amd import _lib as L
    with pytest.raises(L.MclHipError):
        _ctx(grid, n_dir, res_theta, [0.0] * 12 + [500.0])      # 500 + 315 degrees: beyond two turns
    with pytest.raises(L.MclHipError):
        _ctx(grid, n_dir, res_theta, [0.0] * 12 + [-358.0])     # its sector reaches below -2pi


def test_map_without_obstacles_and_all_obstacles():
    grid = np.full((16, 16), 255, np.uint8)
    ctx = _ctx(grid, 4, 90)
    ctx.grid_init_belief()
    ctx.grid_predict(1.5, 0.0, G.DT, ALPHA)
    ctx.grid_update([5.0] * 13)
    pk = ctx.grid_get_belief()[0]
    ref_pred = G.predict(G.initial_belief(grid, 4), grid, (1.5, 0), 4)
    want = G.update(ref_pred, grid, [5.0] * 13, 4)[0]
    assert np.all(np.abs(pk.astype(np.float64) - want) <= 1.3e-7 * np.abs(want))
    wall = np.zeros((16, 16), np.uint8)
    ctx2 = _ctx(wall, 4, 90)
    ctx2.grid_predict(1.5, 0.0, G.DT, ALPHA)
    assert not ctx2.grid_get_belief(False, True)[1].any()


def test_full_size_properties():
    """2048^2 cells x 72 headings (the large configuration): properties that need no oracle."""
    from slam_amd import workload as W
    n, n_dir = 2048, 72
    grid = W.synthetic_map(n)
    ctx = _ctx(grid, n_dir, 5)
    ctx.grid_init_belief()
    ctx.grid_predict(1.5, 0.0, G.DT, ALPHA)
    pp = ctx.grid_get_belief(False, True)[1]
    free = grid >= 200
    interior = np.zeros_like(free)
    interior[3:-3, 3:-3] = True
    assert not pp[~(free & interior)].any()
    assert np.all(pp >= 0) and np.all(np.isfinite(pp))
    # the uniform prior predicts the same value wherever the whole 7x7 neighbourhood is free
    from scipy.ndimage import minimum_filter
    open7 = minimum_filter(free.astype(np.uint8), size=7, mode="constant", cval=0).astype(bool) & interior
    v = pp[open7]
    assert len(v) > 1000
    assert np.max(np.abs(v - v[0])) <= 1e-12 * np.abs(v[0]).max()
    total = ctx.grid_update([2.0, 3.1, 5.0, 1.2, 0.8, 4.4, 5.0, 2.2, 2.9, 3.3, 5.0, 1.0, 0.6])
    pk = ctx.grid_get_belief()[0]
    assert total > 0 and np.isfinite(total)
    assert not pk[~free].any()
    assert np.all(pk >= 0)
    assert abs(float(pk.sum(dtype=np.float64)) - 1.0) < 1e-3   # float32 normalisation of 3e8 terms
    # a 32x32 window against the oracle: predicted ranges (exact up to lattice ties, see above) and the update
    i0, j0 = 1000, 700
    win = np.zeros_like(free)
    win[i0:i0 + 32, j0:j0 + 32] = True
    wi, wj = np.nonzero(free & win)
    assert len(wi) > 500
    angles = list(G.SENSOR_ANGLES)
    Pw, R_g, H_g = _stage_ranges(grid, wi, wj, n_dir, 5, angles)
    lut, cell = O.obstacle_lut(grid)
    near = (np.abs(lut[:, 0] - (j0 + 16) * G.RES_X) < 8.0) & (np.abs(lut[:, 1] - (i0 + 16) * G.RES_Y) < 8.0)   # 5 m + the window
    R_o, H_o = O.sector_query_batch(Pw, lut[near], cell[near], angles)
    R_o, H_o = R_o.reshape(R_g.shape), H_o.reshape(R_g.shape)
    n_ties = _assert_only_lattice_ties(grid, wi, wj, 5, angles, R_g, H_g, R_o, H_o)
    print("2048^2 x 72, 32x32 window: %d of %d ranges differ from the oracle, all exact lattice ties" % (n_ties, R_o.size))
    z_w = [2.0, 3.1, 5.0, 1.2, 0.8, 4.4, 5.0, 2.2, 2.9, 3.3, 5.0, 1.0, 0.6]
    for d in (0, 17, 71):
        q, _ = G.cell_likelihood(z_w, R_g[:, d, :])
        raw = (pp[wi, wj, d] * q).astype(np.float32)                     # robot.py:300
        assert np.array_equal(pk[wi, wj, d], raw / np.float32(total)), d    # pk /= pk.sum() in float32 (robot.py:308)
    # predict is linear in the belief
    ctx.grid_predict(1.5, 0.5, G.DT, ALPHA)
    a = ctx.grid_get_belief(False, True)[1][512:520, 512:520].copy()
    ctx.grid_set_belief(pk * np.float32(0.5))
    ctx.grid_predict(1.5, 0.5, G.DT, ALPHA)
    b = ctx.grid_get_belief(False, True)[1][512:520, 512:520]
    assert np.allclose(b, 0.5 * a, rtol=1e-12, atol=0)


@pytest.mark.parametrize("n_dir,n_beams", [(64, 16), (1, 1), (128, 8)])
def test_table_limits(n_dir, n_beams):
    """The largest (heading, beam) table the update takes -- 1024 entries, as 64 x 16 and as 128 x 8 -- and the
    smallest; one entry more is refused."""
    from slam_amd import _lib as L
    n = 16
    grid = _small_map(n, n_dir)
    res_theta = 360.0 / n_dir
    angles = [k * 360.0 / n_beams + 1.3 for k in range(n_beams)]
    ctx = _ctx(grid, n_dir, res_theta, angles)
    rng = np.random.RandomState(n_dir)
    pred = rng.rand(n, n, n_dir) ** 2
    pred[grid < 200] = 0
    z_meas = list(rng.uniform(0.2, 5.0, n_beams))
    ii, jj, Z = G.predicted_ranges(grid, n_dir, angles=angles, res_theta=res_theta)
    want, total, _ = G.update(pred, grid, z_meas, n_dir, angles=angles, Z=Z)
    ctx.grid_set_prediction(pred)
    got_total = ctx.grid_update(z_meas)
    got = ctx.grid_get_belief()[0]
    assert got_total == total
    assert np.array_equal(got, want), "%d cells differ" % np.sum(got != want)
    if n_dir == 64:
        sp = O.SensorParams()
        with pytest.raises(L.MclHipError):
            ctx.grid_set_sensor(angles + [359.0], n_dir, res_theta, sp.z_max, sp.sigma_hit, sp.lambda_short,
                                [sp.w_hit, sp.w_short, sp.w_max, sp.w_rand], HALF)
