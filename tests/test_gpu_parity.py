"""Parity of the HIP path (called through the C ABI) with the oracle and with the
golden vectors captured from the reference.  Needs an MI355X: `-m gpu`.

Bars: bit-exact for ranges (the float64 min/sqrt chain is exact), hit cells,
free-cell flags, the normalisation sum and resample indices; 1e-12 relative for
float64 poses and weights (device sin/cos/exp differ from numpy's by ulps);
1e-5 relative in float32 state mode.
"""
import os

import numpy as np
import pytest

from oracle import mcl_oracle as O
from util import GOLDEN, load, rel_err, steps

pytestmark = pytest.mark.gpu

ALPHA = [4, 2, 4, 4, 4, 2]
HALF = 5 * np.pi / 180.0


def _lib():
    from slam_amd import _lib
    return _lib


def _sensor(ctx, angles, sp=None):
    sp = sp or O.SensorParams()
    ctx.set_sensor(angles, sp.z_max, sp.sigma_hit, sp.lambda_short, [sp.w_hit, sp.w_short, sp.w_max, sp.w_rand], HALF)


def _grid(name, g):
    return np.load(os.path.join(GOLDEN, "grid_map_g.npy")) if name.startswith("demo") else g["map"]


def _ctx(name, g, dtype=None):
    L = _lib()
    ctx = L.Context(0, L.F64 if dtype is None else dtype)
    ctx.set_map(_grid(name, g), O.RES_X, O.RES_Y)
    _sensor(ctx, list(g["sensor_angles"]))
    return ctx


CASES = ["demo_seed0", "demo_seed1", "demo_seed2", "synth128_b13", "synth256_b180", "synth256_b360"]


@pytest.mark.parametrize("name", CASES)
def test_sector_query_bit_exact(name):
    g = load(name)
    ctx = _ctx(name, g)
    for s in steps(g):
        R, Hc = ctx.raycast(s["X_motion"])
        assert np.array_equal(R, s["ranges"]), "ranges differ in %d places" % np.sum(R != s["ranges"])
        assert np.array_equal(Hc.astype(np.int64), s["hits"])
        Rg, Hg = ctx.raycast(s["x_gt_out"])
        assert np.array_equal(Rg[0], s["z_meas"])
        assert np.array_equal(Hg[0].astype(np.int64), s["hits_gt"])


def test_sector_query_reference_script_answers():
    """Scenarios of the reference's own test_robot.py: every beam it checks reads < 4 m."""
    g = load("test_robot")
    L = _lib()
    for ci, beam, ok in g["checks"]:
        ctx = L.Context(0, L.F64)
        ctx.set_map(g["maps"][ci], O.RES_X, O.RES_Y)
        _sensor(ctx, list(g["sensor_angles"]))
        R, _ = ctx.raycast(g["poses"][ci])
        assert (R[0, beam] < 4) == bool(ok)
        assert np.allclose(R[0], g["ranges_f32path"][ci], rtol=1e-5, atol=1e-5)
        ctx.close()


def test_sector_query_edges():
    """Empty map, pose outside the map, NaN pose, headings far outside one turn."""
    L = _lib()
    ctx = L.Context(0, L.F64)
    grid = np.full((40, 56), 255, np.uint8)
    ctx.set_map(grid, O.RES_X, O.RES_Y)
    angles = [0, 15, 30, 45, 60, 75, 90, 345, 330, 315, 300, 285, 270]
    _sensor(ctx, angles)
    poses = np.array([[1.0, 1.0, 0.3], [-50.0, 3.0, 0.0], [np.nan, 1.0, 0.0], [1.0, 1.0, np.nan], [1e9, 1e9, 1.0]])
    R, Hc = ctx.raycast(poses)
    assert np.all(R == 5.0) and np.all(Hc == -1)
    grid[10, 20] = 0
    grid[30, 5] = 0
    grid[0, 0] = 0
    grid[39, 55] = 0
    ctx.set_map(grid, O.RES_X, O.RES_Y)
    lut, cell = O.obstacle_lut(grid)
    rng = np.random.RandomState(1)
    poses = np.stack([rng.uniform(-1, 9, 300), rng.uniform(-1, 7, 300), rng.uniform(-15, 15, 300)], 1)
    R, Hc = ctx.raycast(poses)
    Ro, Ho = O.sector_query_batch(poses, lut, cell, angles)
    assert np.array_equal(R, Ro) and np.array_equal(Hc.astype(np.int64), Ho)


def test_motion_stage():
    g = load("motion_only")
    L = _lib()
    ctx = L.Context(0, L.F64)
    ctx.set_map(np.load(os.path.join(GOLDEN, "grid_map_g.npy")), O.RES_X, O.RES_Y)
    for c, ut in enumerate(g["controls"]):
        X, pm = ctx.motion(g["X"], ut[0], ut[1], 0.1, g["alpha"], g["noise"][c])
        assert rel_err(X, g["X_out"][c]) < 1e-12
        assert np.array_equal(pm, g["pm"][c])
    Z, pz = ctx.motion(g["X"], 1.5, 0.0, 0.1, g["alpha"], np.zeros((len(g["X"]), 3, 12), np.int16))
    assert rel_err(Z, g["X_out_zero_noise"]) < 1e-12 and np.array_equal(pz, g["pm_zero_noise"])
    # noise=False path
    Xn, _ = ctx.motion(g["X"], 1.5, 0.5, 0.1, g["alpha"], None)
    Xo, _ = O.motion(g["X"], (1.5, 0.5), 0.1, np.load(os.path.join(GOLDEN, "grid_map_g.npy")), None)
    assert rel_err(Xn, Xo) < 1e-12


def test_compute_px_lattice():
    g = load("compute_px")
    L = _lib()
    ctx = L.Context(0, L.F64)
    sp = O.SensorParams(z_max=float(g["z_max"]))
    _sensor(ctx, [0.0], sp)
    p = ctx.compute_px(g["z"], g["z_true"])
    assert rel_err(p, g["p"]) < 1e-12


@pytest.mark.parametrize("name", CASES)
def test_sample_normalise_resample_f64(name):
    g = load(name)
    ctx = _ctx(name, g)
    for s in steps(g):
        ctx.set_particles(s["X_in"])
        ctx.sample(s["ut"][0], s["ut"][1], float(s["dt"]), ALPHA, s["z_meas"], s["noise"])
        X, w, logw = ctx.get_particles(True, True, True)
        assert rel_err(X, s["X_motion"]) < 1e-12
        ok = (s["w_raw"] > 1e-290) | (s["w_raw"] == 0)
        assert rel_err(w[ok], s["w_raw"][ok]) < 1e-11
        assert np.array_equal(w == 0, s["w_raw"] == 0) or np.all(s["w_raw"][w == 0] < 1e-290)
        pos = s["w_raw"] > 1e-290
        assert np.allclose(logw[pos], np.log(s["w_raw"][pos]), rtol=1e-12, atol=1e-11)
        assert np.all(np.isneginf(logw[s["pm"] == 0]))
        # normalisation: the device sum must equal numpy's sum of the device's own weights, bit for bit
        total = ctx.normalise(0)
        assert total == np.sum(w)
        if not bool(s["resample_ok"]):
            continue
        # resampling contract: identical normalised weights + identical r -> identical indices
        for algo in (0, 1):
            ctx.set_particles(s["X_motion"], s["w_raw"])
            assert ctx.normalise(0) == s["w_sum"]
            idx = ctx.resample(float(s["r"]), algo)
            assert np.array_equal(idx, s["idx"])
            X2, w2, _ = ctx.get_particles()
            assert np.array_equal(X2, s["X_out"]) and np.array_equal(w2, s["w_out"])


def test_resample_only_cases_and_large():
    g = load("resample_only")
    L = _lib()
    ctx = L.Context(0, L.F64)
    for c in range(int(g["n_cases"])):
        w = g["c%d_w" % c]
        for algo in (0, 1):
            ctx.set_particles(np.zeros((len(w), 3)), w)
            ctx.normalise(0)
            idx = ctx.resample(float(g["c%d_r" % c]), algo)
            assert np.array_equal(idx, g["c%d_idx" % c]), (c, algo)
            assert np.array_equal(ctx.get_particles()[1], g["c%d_w_out" % c])
        ctx.set_particles(np.zeros((len(w), 3)), w)
        _, idx = ctx.normalise_resample(float(g["c%d_r" % c]), 0)
        assert np.array_equal(idx, g["c%d_idx" % c]), (c, "fused")
        assert np.array_equal(ctx.get_particles()[1], g["c%d_w_out" % c])
    # sizes of the benchmark and weight shapes that stress the order-exact scan
    rng = np.random.RandomState(5)
    for M, kind in [(1 << 20, "rand"), (1 << 20, "peaky"), ((1 << 20) + 12345, "zeros"), (300001, "ties"),
                    (1 << 18, "tiny"), (70000, "equal"), (1 << 20, "copies")]:
        if kind == "rand":
            w = rng.rand(M)
        elif kind == "peaky":
            w = rng.rand(M) ** 60
        elif kind == "zeros":
            w = rng.rand(M) * (rng.rand(M) < 0.2)
            w[:5000] = 0
        elif kind == "ties":
            w = np.ldexp(rng.randint(1, 1 << 20, M).astype(np.float64), -30)  # many exact half-way cases
        elif kind == "tiny":
            w = np.exp(-rng.rand(M) * 700)
        elif kind == "equal":
            w = np.ones(M)
        else:
            w = np.repeat(rng.rand(M // 1024) ** 8, 1024)
        r = rng.rand() / M
        ctx.set_particles(np.zeros((M, 3)), w)
        total = ctx.normalise(0)
        assert total == np.sum(w)
        idx = ctx.resample(r, 1)
        ref = O.resample_indices(w / np.sum(w), r, clamp=True)
        assert np.array_equal(idx, ref), (M, kind, int(np.sum(idx != ref)))
        w_out = ctx.get_particles(False, True, False)[1]
        # the same through the fused update's tail (six kernels; the block that finishes last replays the irregular
        # records of every block -- thousands of them for these shapes, from all over the grid)
        ctx.set_particles(np.zeros((M, 3)), w)
        total_f, idx_f = ctx.normalise_resample(r, 0)
        assert total_f == total and np.array_equal(idx_f, ref), (M, kind, int(np.sum(idx_f != ref)))
        assert np.array_equal(ctx.get_particles(False, True, False)[1], w_out)
    # all-zero weights: NaN after normalisation, every index 0
    ctx.set_particles(np.zeros((1000, 3)), np.zeros(1000))
    ctx.normalise(0)
    for algo in (0, 1):
        ctx.set_particles(np.zeros((1000, 3)), np.zeros(1000))
        ctx.normalise(0)
        assert np.all(ctx.resample(0.0004, algo) == 0)


def test_log_weight_mode_matches_product_mode():
    g = load("synth128_b13")
    s = steps(g)[0]
    ctx = _ctx("synth128_b13", g)
    out = []
    for mode in (0, 1):
        ctx.set_particles(s["X_in"])
        ctx.sample(s["ut"][0], s["ut"][1], float(s["dt"]), ALPHA, s["z_meas"], s["noise"])
        ctx.normalise(mode)
        out.append(ctx.get_particles()[1])
    assert np.allclose(out[0], out[1], rtol=1e-10, atol=1e-300)


@pytest.mark.parametrize("name", ["demo_seed0", "synth256_b180"])
def test_f32_state_mode(name):
    """float32 particle state: poses and log-weights within 1e-5 relative of the float64
    reference values; ranges / hit cells are exact for the poses the device holds."""
    g = load(name)
    L = _lib()
    ctx = _ctx(name, g, L.F32)
    grid = _grid(name, g)
    lut, cell = O.obstacle_lut(grid)
    s = steps(g)[0]
    ctx.set_particles(s["X_in"])
    ctx.sample(s["ut"][0], s["ut"][1], float(s["dt"]), ALPHA, s["z_meas"], s["noise"])
    X, w, logw = ctx.get_particles(True, True, True)
    assert np.max(np.abs(X - s["X_motion"]) / np.maximum(1.0, np.abs(s["X_motion"]))) < 1e-5
    # stage contract on the device's own float32 poses
    R, Hc = ctx.raycast(X)
    Ro, Ho = O.sector_query_batch(X, lut, cell, list(g["sensor_angles"]))
    assert np.array_equal(R, Ro) and np.array_equal(Hc.astype(np.int64), Ho)
    wo, lwo, _ = O.weights_from_ranges(Ro, s["z_meas"], O.free_cell(X[:, 0], X[:, 1], grid), O.SensorParams())
    fin = np.isfinite(lwo)
    assert np.array_equal(np.isfinite(logw), fin)
    assert np.max(np.abs(logw[fin] - lwo[fin]) / np.maximum(1.0, np.abs(lwo[fin]))) < 1e-5
    # the log-weight-only pass (what the fused step and bench.py run: the float32 beam mixture)
    ctx.set_particles(s["X_in"])
    ctx.shard_sample(s["ut"][0], s["ut"][1], float(s["dt"]), ALPHA, s["z_meas"], s["noise"], 0, L.W_LOG)
    X2, _, logw2 = ctx.get_particles(True, False, True)
    assert np.array_equal(X2, X)
    assert np.array_equal(np.isfinite(logw2), fin)
    err = np.abs(logw2[fin] - lwo[fin]) / np.maximum(1.0, np.abs(lwo[fin]))
    assert err.max() < 1e-5, err.max()
    print("%s: float32 log-weight error, product pass %.2e, log-only pass %.2e" %
          (name, np.max(np.abs(logw[fin] - lwo[fin]) / np.maximum(1.0, np.abs(lwo[fin]))), err.max()))
    # a measurement the float32 mixture does not take (negative range): the general path answers, same bar
    zneg = np.array(s["z_meas"], np.float64).copy()
    zneg[0] = -0.25
    ctx.set_particles(s["X_in"])
    ctx.shard_sample(s["ut"][0], s["ut"][1], float(s["dt"]), ALPHA, zneg, s["noise"], 0, L.W_LOG)
    _, _, logw3 = ctx.get_particles(False, False, True)
    _, lwn, _ = O.weights_from_ranges(Ro, zneg, O.free_cell(X[:, 0], X[:, 1], grid), O.SensorParams())
    assert np.max(np.abs(logw3[fin] - lwn[fin]) / np.maximum(1.0, np.abs(lwn[fin]))) < 1e-5


@pytest.mark.parametrize("every", ["0", "1"])
@pytest.mark.parametrize("beams", ["even180", "even360", "reference13", "uneven", "wide_repeated", "narrow", "single"])
def test_sector_intervals_against_the_oracle(beams, every, monkeypatch):
    """The float32 instantiation files a cell under the sector interval its bearing falls in (one minimum per interval,
    a beam's range = the smallest of the intervals its sector spans) when no beam angle + heading leaves the range in
    which the reference recognises a turn (mc_localization.py:148-149); other headings go beam by beam, bearings within
    rounding of a sector edge are decided in float64.  Headings on both sides of the two
    limits, negative ones and ones past a full turn, evenly and unevenly spaced beam sets (overlapping sectors, disjoint
    sectors, sectors that share an edge): log-weights against the oracle's, which follows mc_localization.py:116-171.
    every = "1": evenly spaced beams whose spacing divides the sector (even180, even360) have the headings outside that
    range served by a second pass of the interval kernel (cells at the seam of the bearings asked beam by beam, beams
    two or more turns away seeing nothing) instead of entry by entry; the other tables do not have that pass."""
    from slam_amd import workload as W
    L = _lib()
    monkeypatch.setenv("MCLHIP_EVERY_HEADING", every)       # (read when a context is created)
    monkeypatch.setenv("MCLHIP_SPLIT", "1")
    grid = W.synthetic_map(256)
    if beams == "even180":
        angles = W.beam_angles(180)
    elif beams == "even360":
        angles = W.beam_angles(360)
    elif beams == "reference13":
        angles = [0, 15, 30, 45, 60, 75, 90, 345, 330, 315, 300, 285, 270]           # mc_localization.py:45
    elif beams == "uneven":
        angles = [0.0, 3.0, 4.5, 10.0, 20.0, 20.5, 77.7, 180.0, 181.0, 190.0, 200.0, 269.9, 275.0, 359.0]   # 10 and 20: a shared edge
    elif beams == "wide_repeated":
        angles = [0.0, 0.0, 10.0, 100.0, 100.0, 100.0, 185.0, 359.0, 40.0]           # 30-degree sectors, beams given twice
    elif beams == "narrow":
        angles = [k * 7.5 for k in range(48)]                                        # 1-degree sectors, gaps between them
    else:
        angles = [123.0]
    half = {"wide_repeated": 30.0, "narrow": 1.0}.get(beams, 5.0) * np.pi / 180.0
    rng = np.random.RandomState(5)
    M = 1500
    X = W.uniform_particles(grid, M, seed=3)
    th = rng.uniform(0.0, 2 * np.pi, M)
    arad = np.asarray(angles, float) * np.pi / 180.0
    lim_hi = 4 * np.pi - half - arad.max() - 1e-3             # the interval path serves headings in [lim_lo, lim_hi]
    lim_lo = -2 * np.pi + half - arad.min() + 1e-3
    th[:80] = lim_hi + rng.uniform(-2e-3, 2e-3, 80)           # around the upper limit of the interval path
    th[80:160] = lim_lo + rng.uniform(-2e-3, 2e-3, 80)        # around the lower limit
    th[160:200] = 2 * np.pi - 2 * half + rng.uniform(-2e-3, 2e-3, 40)   # where the highest sector starts to pass two turns
    th[200:400] = rng.uniform(-1e-3, 1e-3, 200)               # around zero
    th[400:500] = rng.uniform(-7.0, 0.0, 100)                 # negative, some below one turn
    th[500:600] = rng.uniform(2 * np.pi, 13.5, 100)           # past one and two turns
    th[600:700] = np.round(rng.uniform(0, 72, 100)) * (5 * np.pi / 180.0)   # sector edges on lattice directions
    X[:, 2] = th
    X = X.astype(np.float32).astype(np.float64)               # what a float32 state holds
    ctx = L.Context(0, L.F32)
    ctx.set_map(grid, W.RES, W.RES)
    w4 = np.array([0.97, 0.005, 0.020, 0.005]); w4 /= w4.sum()
    ctx.set_sensor(angles, W.Z_MAX, 0.15, 1.0, w4, half)
    z_meas = ctx.raycast(np.array([3.0, 4.5, 0.3]), want_hits=False)[0][0]
    ctx.set_particles(X)
    zero = np.zeros((M, 3, 12), np.int16)
    ctx.shard_sample(0.0, 1e-12, 0.1, ALPHA, z_meas, zero, 0, L.W_LOG)     # no motion: the poses stay what they are
    Xd, _, logw = ctx.get_particles(True, False, True)
    lut, cell = O.obstacle_lut(grid)
    Ro, _ = O.sector_query_batch(Xd, lut, cell, angles, half_angle=half)
    _, lwo, _ = O.weights_from_ranges(Ro, z_meas, O.free_cell(Xd[:, 0], Xd[:, 1], grid), O.SensorParams())
    fin = np.isfinite(lwo)
    assert np.array_equal(np.isfinite(logw), fin) and fin.sum() > 0.8 * M
    err = np.abs(logw[fin] - lwo[fin]) / np.maximum(1.0, np.abs(lwo[fin]))
    # an obstacle centre whose bearing sits on a sector edge to the last bit is in or out by atan2's last bit (libm's
    # against the device's): lattice headings produce such ties, everything else must agree
    tie = np.zeros(M, bool); tie[600:700] = True
    assert err[~tie[fin]].max() < 1e-5, (beams, err[~tie[fin]].max())
    assert np.mean(err[tie[fin]] < 1e-5) > 0.9
    first_left, then_left = ctx.timing_get(7)[0], ctx.timing_get(6)[0]
    if beams in ("even180", "even360"):
        outside = int(np.sum((Xd[:, 2] > lim_hi) | (Xd[:, 2] < lim_lo)))
        assert first_left >= outside - 80                      # (within 1e-3 of a limit the kernel may draw the line either way)
        assert (then_left < first_left // 4) if every == "1" else (then_left == first_left)
    ctx.close()


def test_augmented_mcl_device_noise_in_the_float32_mixture():
    """Float32 state: the device-drawn range noise of augmented MCL goes through the interval-only sector query (noise
    drawn inside its float32 mixture, raw weights as exponentials of the log-weights).  Its noise is a stream of its
    own, so what is compared with the float64 path -- the general kernel, factor by factor -- is the distribution: per
    particle the two log-weights differ by noise alone, with no bias; on a map without obstacles the moments are
    those numpy gives for log p over n ~ N(0, 1) (mc_localization2.py:166-173)."""
    from slam_amd import workload as W
    L = _lib()
    sp = O.SensorParams()
    angles = [0, 15, 30, 45, 60, 75, 90, 345, 330, 315, 300, 285, 270]           # mc_localization2.py's beam set
    grid = W.synthetic_map(256, target_occupancy=0.25)       # dense: few beams read z_max, where half the noisy ranges are out of range
    M = 1 << 17
    X = W.uniform_particles(grid, M, seed=5).astype(np.float32).astype(np.float64)
    got = {}
    for dtype in (L.F32, L.F64):
        ctx = L.Context(0, dtype)
        ctx.set_map(grid, W.RES, W.RES)
        _sensor(ctx, angles)
        z_meas = ctx.raycast(np.array([5.0, 7.5, 0.4]), want_hits=False)[0][0]
        ctx.set_particles(X)
        w_avg = ctx.sample_augmented_device(0.0, 0.0, 0.1, ALPHA, z_meas, 777, 0.15)     # no motion
        Xd, w, logw = ctx.get_particles(True, True, True)
        assert np.array_equal(Xd, X)
        assert np.isclose(w_avg, np.mean(w), rtol=1e-9)
        fin = np.isfinite(logw)
        assert np.allclose(w[fin], np.exp(logw[fin]), rtol=1e-6) and np.all(w[~fin] == 0.0)
        got[dtype] = (logw, ctx.query_kernel_name())
        ctx.close()
    # the float32 state took the interval-only kernel with the noise in it (template argument 6), the float64 state the general one
    # (... over equal intervals: the reference's beam set has every sector edge on a multiple of five degrees)
    assert got[L.F32][1] == "raycast_kernel<float,false,false,false,4,true,true,true,false,192,-1>", got[L.F32][1]
    a, b = got[L.F32][0], got[L.F64][0]
    both = np.isfinite(a) & np.isfinite(b)
    # (a noisy range beyond z_max or below zero has likelihood zero: which particles that hits is the noise's doing)
    fa, fb = np.isfinite(a).mean(), np.isfinite(b).mean()
    assert abs(fa - fb) < 6 * np.sqrt(2 * fb * (1 - fb) / M) and both.sum() > 0.2 * M, (fa, fb)
    d = a[both] - b[both]
    assert d.std() > 0.05                                                        # different noise ...
    assert abs(d.mean()) < 6 * d.std() / np.sqrt(both.sum())                     # ... the same distribution
    assert abs(a[both].std() / b[both].std() - 1) < 0.02
    # without obstacles: min(z_max, z_max + 0.15 n) on every beam
    rng = np.random.RandomState(2)
    M2, B = 1 << 18, 4
    free = np.full((100, 100), 255, np.uint8)
    ctx = L.Context(0, L.F32)
    ctx.set_map(free, O.RES_X, O.RES_Y)
    _sensor(ctx, [0, 90, 180, 270])
    X2 = np.stack([rng.uniform(1, 5, M2), rng.uniform(1, 5, M2), rng.uniform(0, 6, M2)], 1)
    ctx.set_particles(X2)
    z_meas = np.array([5.0, 4.9, 4.7, 3.0])
    ctx.sample_augmented_device(0.0, 0.0, 0.1, ALPHA, z_meas, 4242, 0.15)
    logw = ctx.get_particles(True, True, True)[2]
    n = rng.randn(2_000_000, B)
    zp = np.minimum(sp.z_max, sp.z_max + 0.15 * n)
    want = np.log(O.beam_px(zp, z_meas[None, :], sp)).sum(1)
    assert abs(logw.mean() - want.mean()) < 6 * want.std() / np.sqrt(M2)
    assert abs(logw.std() / want.std() - 1) < 0.02
    assert abs(np.corrcoef(logw[:-1], logw[1:])[0, 1]) < 5 / np.sqrt(M2)
    ctx.close()


def test_dropin_class_follows_reference_run():
    """The drop-in class driven like the reference's __main__ (mc_localization.py:303-331) under
    the same np.random seed: same initial cloud, same indices, same trajectory."""
    from slam_amd.mc_localization import MCLocalization, res_x, res_y
    g = load("demo_seed0")
    grid = np.load(os.path.join(GOLDEN, "grid_map_g.npy"))
    np.random.seed(0)
    x0 = np.array([20 * res_x, 30 * res_y, 0], dtype=np.float32)
    mcl = MCLocalization(x0, grid)
    mcl.initialize(x0)
    for t, s in enumerate(steps(g)):
        assert np.array_equal(mcl.X, s["X_in"]) if t == 0 else rel_err(mcl.X, s["X_in"]) < 1e-12
        mcl.sample([1.5, 0], 0.1)
        assert np.array_equal(np.asarray(mcl.x_gt), s["x_gt_out"])
        assert np.array_equal(np.asarray(mcl.sensor_ranges), s["z_meas"])
        mcl.resample()
        assert np.array_equal(mcl.last_idx, s["idx"])
        assert rel_err(mcl.X, s["X_out"]) < 1e-12
        assert rel_err(mcl.w, s["w_out"]) < 1e-11
    rs = mcl.range_sensor
    gp = load("compute_px")
    assert abs(rs.compute_px(1.0, 1.2) - O.beam_px(1.0, 1.2, O.SensorParams())) < 1e-15
    assert rel_err(rs.compute_px(gp["z"], gp["z_true"]), gp["p"]) < 1e-12


def test_device_noise_statistics():
    """The on-device generator (throughput runs) is not the reference's stream; its draws must have
    the distribution of Table 5.4: zero mean, standard deviation b/6*sqrt(12*var(U{-1000..1000})/1e6)."""
    L = _lib()
    ctx = L.Context(0, L.F64)
    grid = np.full((64, 64), 255, np.uint8)
    ctx.set_map(grid, O.RES_X, O.RES_Y)
    _sensor(ctx, [0.0, 90.0])
    M = 200000
    X0 = np.tile(np.array([[4.0, 4.0, 0.0]]), (M, 1))
    ctx.set_particles(X0)
    ctx.sample(1.5, 0.0, 0.1, ALPHA, [5.0, 5.0], None, seed=1234)
    X, _, _ = ctx.get_particles()
    th = X[:, 2]  # theta' = (w_hat + g_hat) * dt
    sd_unit = np.sqrt(12 * (2001 ** 2 - 1) / 12.0) * 0.001
    sd_expected = 0.1 * np.hypot(1.0 * sd_unit, 1.0 * sd_unit)
    assert abs(np.mean(th)) < 5 * sd_expected / np.sqrt(M)
    assert abs(np.std(th) / sd_expected - 1) < 0.02
    ctx.set_particles(X0)
    ctx.sample(1.5, 0.0, 0.1, ALPHA, [5.0, 5.0], None, seed=1234)
    assert np.array_equal(ctx.get_particles()[0], X)  # same seed, same draws


@pytest.mark.parametrize("name", ["amcl_seed0", "amcl_seed4"])
def test_augmented_mcl_stages(name):
    """mc_localization2.py through mclhip_sample_augmented / mclhip_resample_augmented."""
    g = load(name)
    L = _lib()
    ctx = L.Context(0, L.F64)
    grid = np.load(os.path.join(GOLDEN, "grid_map_g.npy"))
    ctx.set_map(grid, O.RES_X, O.RES_Y)
    _sensor(ctx, list(g["sensor_angles"]))
    for s in steps(g):
        ctx.set_particles(s["X_in"])
        w_avg = ctx.sample_augmented(1.5, 0.0, 0.1, ALPHA, s["z_meas"], s["noise"], s["range_noise"])
        X, w, _ = ctx.get_particles()
        assert rel_err(X, s["X_motion"]) < 1e-12
        assert np.array_equal(w == 0, s["w_raw"] == 0)
        assert rel_err(w, s["w_raw"]) < 1e-11
        assert abs(w_avg - s["w_avg"]) <= 1e-11 * abs(s["w_avg"])
        assert w_avg == O.sequential_mean(w)          # the running mean is the exact sequential chain
        # resampling contract on the reference's own weights
        ctx.set_particles(s["X_motion"], s["w_raw"])
        assert ctx.normalise(0) == s["w_sum"]
        ctx.resample_augmented(float(s["r"]), s["inj_rows"], s["inj_poses"])
        X2, w2, _ = ctx.get_particles()
        assert np.array_equal(X2, s["X_out"]) and np.array_equal(w2, s["w_out"])


def test_augmented_dropin_class_follows_reference_run():
    from slam_amd.mc_localization2 import MCLocalization, res_x, res_y
    g = load("amcl_seed0")
    grid = np.load(os.path.join(GOLDEN, "grid_map_g.npy"))
    np.random.seed(0)
    x0 = np.array([20 * res_x, 30 * res_y, 0], dtype=np.float32)
    mcl = MCLocalization(x0, grid, M=300)
    mcl.initialize(x0)
    for t, s in enumerate(steps(g)[:4]):   # later steps have their slow mean overridden in the fixture
        assert rel_err(mcl.X, s["X_in"]) < 1e-12
        mcl.sample([1.5, 0], 0.1)
        assert np.array_equal(np.asarray(mcl.sensor_ranges), s["z_meas"])
        assert abs(mcl.w_slow - s["w_slow"]) <= 1e-11 * abs(s["w_slow"])
        assert abs(mcl.w_fast - s["w_fast"]) <= 1e-11 * abs(s["w_fast"])
        mcl.resample()
        assert np.array_equal(mcl.last_injected, s["inj_rows"])
        assert rel_err(mcl.X, s["X_out"]) < 1e-12


def test_device_side_initialisation_has_the_reference_distribution():
    """mclhip_init_particles (mc_localization.py:173-219 on the device): every pose rounds to a free
    cell; the uniform cloud covers the free space evenly; the normal cloud has the reference's
    moments.  Statistical parity only: the draws are the device generator's, not numpy's."""
    from slam_amd.mc_localization import MCLocalization, max_x, max_y
    grid = np.load(os.path.join(GOLDEN, "grid_map_g.npy"))
    M = 200000
    mcl = MCLocalization(np.array([3.0, 4.5, 0.0]), grid, M=M)
    mcl.initialize(on_device=True, seed=5)
    X = mcl.X.copy()
    assert X.shape == (M, 3) and np.all(np.isfinite(X))
    assert np.all(O.free_cell(X[:, 0], X[:, 1], grid) == 1.0)
    assert X[:, 2].min() >= 0 and X[:, 2].max() < 2 * np.pi and abs(X[:, 2].mean() - np.pi) < 0.02
    # even coverage: particle counts per free cell are Poisson with the same mean
    u = np.trunc(X[:, 0] / O.RES_X + 0.5).astype(int)
    v = np.trunc(X[:, 1] / O.RES_Y + 0.5).astype(int)
    cnt = np.zeros(grid.shape, int)
    np.add.at(cnt, (v, u), 1)
    free = grid == 255
    lam = M / free.sum()
    assert abs(cnt[free].mean() - lam) < 1e-9 and abs(cnt[free].var() / lam - 1.0) < 0.1
    assert np.allclose(mcl.w, 1.0 / M)
    # normal cloud on an all-free map: moments of the proposal itself
    mcl2 = MCLocalization(np.array([7.0, 8.0, 0.5]), np.full((100, 100), 255, np.uint8), M=M)
    mcl2.initialize(np.array([7.0, 8.0, 0.5]), on_device=True, seed=6)
    Y = mcl2.X
    inside = (Y[:, 0] > 0) & (Y[:, 1] > 0)
    assert inside.all()
    assert abs(Y[:, 2].mean() - 0.5) < 0.03 and abs(Y[:, 2].std() - np.pi) < 0.03
    assert abs(np.median(Y[:, 0]) - 7.0) < 0.03 and abs(np.median(Y[:, 1]) - 8.0) < 0.03
    # and it feeds the filter like any other cloud
    mcl.sample([1.5, 0], 0.1)
    mcl.resample()
    assert np.isfinite(mcl.w).all()


def test_limits_and_smallest_sizes():
    """The largest beam table (1024), one beam, one particle, a one-cell map; sizes beyond the limits are refused."""
    L = _lib()
    rng = np.random.RandomState(3)
    grid = np.full((48, 64), 255, np.uint8)
    grid[rng.randint(0, 48, 150), rng.randint(0, 64, 150)] = 0
    lut, cell = O.obstacle_lut(grid)
    poses = np.stack([rng.uniform(0, 9.6, 6), rng.uniform(0, 7.2, 6), rng.uniform(-4, 4, 6)], 1)
    for B in (1024, 1, 7):
        angles = [k * 360.0 / B for k in range(B)]
        ctx = L.Context(0, L.F64)
        ctx.set_map(grid, O.RES_X, O.RES_Y)
        _sensor(ctx, angles)
        R, Hc = ctx.raycast(poses)
        Ro, Ho = O.sector_query_batch(poses, lut, cell, angles)
        assert np.array_equal(R, Ro) and np.array_equal(Hc.astype(np.int64), Ho), B
        # one particle through the whole update
        ctx.set_particles(poses[:1])
        noise = rng.randint(-1000, 1001, size=(1, 3, 12)).astype(np.int16)
        ctx.sample(1.5, 0.5, 0.1, ALPHA, R[1], noise)
        Xo, pmo = O.motion(poses[:1], (1.5, 0.5), 0.1, grid, noise)
        X, w, logw = ctx.get_particles(True, True, True)
        assert rel_err(X, Xo) < 1e-12
        total = ctx.normalise(0)
        assert total == np.sum(w)
        assert np.array_equal(ctx.resample(0.25), [0])
        ctx.close()
    ctx = L.Context(0, L.F64)
    with pytest.raises(L.MclHipError):
        _sensor(ctx, [k * 360.0 / 1025 for k in range(1025)])
    with pytest.raises(L.MclHipError):
        ctx.raycast(poses)                                  # nothing set yet
    one = np.zeros((1, 1), np.uint8)
    ctx.set_map(one, O.RES_X, O.RES_Y)
    _sensor(ctx, [0.0, 90.0, 180.0])
    R, Hc = ctx.raycast(np.array([[1.0, 0.0, 0.0], [0.0, 0.0, 0.0]]))
    Ro, Ho = O.sector_query_batch(np.array([[1.0, 0.0, 0.0], [0.0, 0.0, 0.0]]), *O.obstacle_lut(one), [0.0, 90.0, 180.0])
    assert np.array_equal(R, Ro) and np.array_equal(Hc.astype(np.int64), Ho)
    ctx.close()


def test_widest_map():
    """A map as wide as the kernels take (65535 cells; cell coordinates are packed in 16 bits): obstacles and poses
    at the far end, where the float32 cell coordinates have lost most of their fraction bits."""
    L = _lib()
    H, W = 12, 65535
    grid = np.full((H, W), 255, np.uint8)
    rng = np.random.RandomState(8)
    cols = np.concatenate([rng.randint(W - 80, W, 60), rng.randint(32700, 32840, 60), rng.randint(0, 60, 30)])
    grid[rng.randint(0, H, len(cols)), cols] = 0
    lut, cell = O.obstacle_lut(grid)
    angles = [0, 15, 30, 45, 60, 75, 90, 345, 330, 315, 300, 285, 270]
    ctx = L.Context(0, L.F64)
    ctx.set_map(grid, O.RES_X, O.RES_Y)
    _sensor(ctx, angles)
    xs = np.concatenate([rng.uniform((W - 90) * 0.15, (W + 5) * 0.15, 40), rng.uniform(32690 * 0.15, 32850 * 0.15, 40),
                         rng.uniform(-1, 10, 20)])
    poses = np.stack([xs, rng.uniform(-0.5, H * 0.15 + 0.5, len(xs)), rng.uniform(-7, 7, len(xs))], 1)
    R, Hc = ctx.raycast(poses)
    Ro, Ho = O.sector_query_batch(poses, lut, cell, angles)
    assert np.array_equal(R, Ro) and np.array_equal(Hc.astype(np.int64), Ho)
    assert np.any(R < 5.0)
    X, pm = ctx.motion(poses, 1.5, 0.5, 0.1, ALPHA, None)
    Xo, pmo = O.motion(poses, (1.5, 0.5), 0.1, grid, None)
    assert rel_err(X, Xo) < 1e-12 and np.array_equal(pm, pmo)
    with pytest.raises(L.MclHipError):
        ctx.set_map(np.full((2, 65536), 255, np.uint8), O.RES_X, O.RES_Y)
    ctx.close()


@pytest.mark.parametrize("res,z_max,half_deg,angles", [
    ((0.02, 0.02), 5.0, 5.0, [0, 15, 30, 45, 60, 75, 90, 345, 330, 315, 300, 285, 270]),      # 250 cells of reach
    ((0.15, 0.4), 5.0, 5.0, [0, 90, 180, 270, 10, 20]),                                       # oblong cells
    ((0.15, 0.15), 3.0, 1.0, [k * 360.0 / 50 for k in range(50)]),                            # narrow sectors
    ((0.15, 0.15), 5.0, 30.0, [0, 15, 30, 45, 60, 75, 90, 345, 330, 315, 300, 285, 270]),     # overlapping wide sectors
    ((0.15, 0.15), 5.0, 5.0, [0, 0, 90, 90.0, -30, 400, 720, -359.5, 359.5, 180, 180]),       # repeated and unreduced angles
    ((0.15, 0.15), 5.0, 40.0, [k * 0.9 for k in range(400)]),                                 # 90 beams reach across the 0 / 2 pi seam
    ((0.15, 0.15), 4.0, 100.0, [(k * 1.2 + 0.3) % 360.0 for k in range(300)]),                # 167 do, sectors wider than a half turn
])
def test_sector_query_other_geometries(res, z_max, half_deg, angles):
    L = _lib()
    rng = np.random.RandomState(int(half_deg * 10 + len(angles)))
    n = 300 if res[0] < 0.1 else 70
    grid = np.full((n, n), 255, np.uint8)
    grid[rng.randint(0, n, 3 * n), rng.randint(0, n, 3 * n)] = 0
    lut, cell = O.obstacle_lut(grid, res[0], res[1])
    sp = O.SensorParams(z_max=z_max)
    half = half_deg * np.pi / 180.0
    ctx = L.Context(0, L.F64)
    ctx.set_map(grid, res[0], res[1])
    ctx.set_sensor(angles, sp.z_max, sp.sigma_hit, sp.lambda_short, [sp.w_hit, sp.w_short, sp.w_max, sp.w_rand], half)
    poses = np.stack([rng.uniform(-0.5, n * res[0] + 0.5, 120), rng.uniform(-0.5, n * res[1] + 0.5, 120),
                      rng.uniform(-7, 7, 120)], 1)
    R, Hc = ctx.raycast(poses)
    Ro, Ho = O.sector_query_batch(poses, lut, cell, angles, z_max=z_max, half_angle=half)
    assert np.array_equal(R, Ro), int(np.sum(R != Ro))
    assert np.array_equal(Hc.astype(np.int64), Ho)
    assert np.any(R < z_max) and np.any(R == z_max)
    ctx.close()


def test_augmented_mcl_with_device_noise():
    """Augmented MCL with the range noise and the injections drawn on the device: their distributions.
    On a map without obstacles a predicted range is min(z_max, z_max + 0.15 n) (mc_localization2.py:170-173), so
    the mean and the spread of the log-weights are those of log p over n ~ N(0, 1), which numpy gives directly;
    the injected fraction follows p_random and injected poses stand on free cells with weight 1/M."""
    L = _lib()
    rng = np.random.RandomState(2)
    M, B = 1 << 18, 4
    grid = np.full((100, 100), 255, np.uint8)
    grid[40:60, 40:60] = 127                       # neither free nor obstacle
    sp = O.SensorParams()
    ctx = L.Context(0, L.F64)
    ctx.set_map(grid, O.RES_X, O.RES_Y)
    _sensor(ctx, [0, 90, 180, 270])
    X = np.stack([rng.uniform(1, 5, M), rng.uniform(1, 5, M), rng.uniform(0, 6, M)], 1)
    ctx.set_particles(X)
    z_meas = np.array([5.0, 4.9, 4.7, 3.0])
    w_avg = ctx.sample_augmented_device(0.0, 0.0, 0.1, ALPHA, z_meas, 4242, 0.15)     # no motion: every pose stays free
    _, w, logw = ctx.get_particles(True, True, True)
    n = rng.randn(2_000_000, B)
    zp = np.minimum(sp.z_max, sp.z_max + 0.15 * n)
    want = np.log(O.beam_px(zp, z_meas[None, :], sp)).sum(1)
    assert abs(logw.mean() - want.mean()) < 6 * want.std() / np.sqrt(M)
    assert abs(logw.std() / want.std() - 1) < 0.02
    assert np.isclose(w_avg, np.mean(w), rtol=1e-9)
    # independent noise per particle and per beam: neighbouring particles uncorrelated
    assert abs(np.corrcoef(logw[:-1], logw[1:])[0, 1]) < 5 / np.sqrt(M)
    # injections
    total = ctx.normalise(0)
    assert total == np.sum(w)
    n_inj = ctx.resample_augmented_device(0.3 / M, 0.25, 15.0, 15.0, 99)
    assert abs(n_inj / M - 0.25) < 5 * np.sqrt(0.25 * 0.75 / M)
    X2, w2, _ = ctx.get_particles()
    fresh = w2 == 1.0 / M
    assert abs(int(fresh.sum()) - n_inj) <= 2      # a resampled weight may equal 1/M by chance
    u = (X2[fresh, 0] / O.RES_X + 0.5).astype(int)
    v = (X2[fresh, 1] / O.RES_Y + 0.5).astype(int)
    assert np.all(grid[v, u] == 255)
    assert X2[fresh, 0].min() < 1.0 and X2[fresh, 0].max() > 14.0 and X2[fresh, 2].max() > 6.0   # the whole extent, not the cloud's box
    # nothing is injected at p_random 0 or NaN: plain resampling
    ctx.set_particles(X, w)
    ctx.normalise(0)
    idx = ctx.resample(0.3 / M)
    ctx.set_particles(X, w)
    ctx.normalise(0)
    assert ctx.resample_augmented_device(0.3 / M, float("nan"), 15.0, 15.0, 99) == 0
    assert np.array_equal(ctx.get_particles()[0], X[idx])
    ctx.close()


@pytest.mark.parametrize("dtype,mode", [(1, 0), (1, 1), (0, 1)])
def test_fused_step_equals_the_separate_calls(dtype, mode):
    """mclhip_step (six launches for normalisation + order-exact resampling, the last block of one of them replaying
    the irregular records) against mclhip_sample / normalise / resample: same weights, same indices, same poses,
    for particle counts around every tile / chunk boundary and over several steps in a row."""
    L = _lib()
    from slam_amd import workload as W
    grid = W.synthetic_map(256)
    angles = W.beam_angles(24)
    rng = np.random.RandomState(17)
    for M in (1, 63, 500, 2048, 2049, 8191, 8192, 8193, 40000, 1 << 20):
        X = W.uniform_particles(grid, M, seed=M % 1000 + 3)
        a = L.Context(0, dtype)
        b = L.Context(0, dtype)
        for c in (a, b):
            c.set_map(grid, W.RES, W.RES)
            _sensor(c, angles)
            c.set_particles(X)
        z = a.raycast(X[0], want_hits=False)[0][0]
        for t in range(3):
            noise = rng.randint(-1000, 1001, size=(M, 3, 12)).astype(np.int16) if M <= 40000 else None
            r = float(rng.rand()) / M
            a.step(1.5, 0.5 * (t % 2), 0.1, ALPHA, z, r, noise, 50 + t, mode, wait=True)
            b.shard_sample(1.5, 0.5 * (t % 2), 0.1, ALPHA, z, noise, 50 + t, mode)    # same kernels as the fused step's first half
            total = b.normalise(mode)
            idx = b.resample(r)
            assert np.array_equal(a.get_indices(), idx), (M, t)
            Xa, wa, _ = a.get_particles()
            Xb, wb, _ = b.get_particles()
            assert np.array_equal(Xa, Xb) and np.array_equal(wa, wb), (M, t)
            assert np.isfinite(total)
        a.close()
        b.close()


def test_one_kernel_and_two_launch_sector_query_agree_bit_for_bit(monkeypatch):
    """The float32 throughput path runs the sector query as two launches (interval-only kernel + the general kernel over
    what it leaves) or, when most of the cloud lies outside the interval path's heading range, as the one kernel that
    holds both paths (MCLHIP_SPLIT = 1 / 0 force either; read when a context is created); between the two launches the
    interval kernel may make a second pass, over the list, that serves the headings outside the range as well
    (MCLHIP_EVERY_HEADING = 1 / 0; by itself when a sixteenth to half of the cloud was left).  Which of these run is a
    matter of speed only: the log-weights are the same bits."""
    from slam_amd import workload as W
    L = _lib()
    grid = W.synthetic_map(256)
    rng = np.random.RandomState(9)
    M = 20_011
    X = W.uniform_particles(grid, M, seed=5)
    th = rng.uniform(-6.5, 6.5, M)                       # a sixth outside the range (|heading| > 355 degrees and beyond)
    th[:500] = np.round(rng.uniform(0, 72, 500)) * (5 * np.pi / 180.0)   # lattice headings: cells exactly on sector edges
    X[:, 2] = th
    # poses on and next to the corners, edges and centres of the 8 x 8-cell tiles the cell lists are kept for: the
    # two-launch path reads only the part of a tile's list that a pose at that distance from the tile's centre can
    # reach (the one-kernel path reads all of it)
    k = np.arange(2000)
    X[1000:3000, 0] = (8 * rng.randint(2, 30, 2000) + np.array([0.0, 4.0, 8.0, 7.999])[k % 4]) * W.RES + rng.choice([0.0, 1e-6, -1e-6], 2000)
    X[1000:3000, 1] = (8 * rng.randint(2, 30, 2000) + np.array([0.0, 4.0, 8.0, 7.999])[(k // 4) % 4]) * W.RES + rng.choice([0.0, 1e-6, -1e-6], 2000)
    X = X.astype(np.float32).astype(np.float64)
    w4 = np.array([0.97, 0.005, 0.020, 0.005]); w4 /= w4.sum()
    zero = np.zeros((M, 3, 12), np.int16)
    got = {}
    for mode in ("0", "1", "1+every", "auto"):
        monkeypatch.delenv("MCLHIP_EVERY_HEADING", raising=False)
        if mode == "auto":
            monkeypatch.delenv("MCLHIP_SPLIT", raising=False)
        else:
            monkeypatch.setenv("MCLHIP_SPLIT", mode[0])
            monkeypatch.setenv("MCLHIP_EVERY_HEADING", "1" if mode.endswith("every") else "0")
        ctx = L.Context(0, L.F32)
        ctx.set_map(grid, W.RES, W.RES)
        ctx.set_sensor(W.beam_angles(180), W.Z_MAX, 0.15, 1.0, w4, 5 * np.pi / 180.0)
        z_meas = ctx.raycast(np.array([3.0, 4.5, 0.3]), want_hits=False)[0][0]
        ctx.set_particles(X)
        for _ in range(3):                                # (auto: the second and third query know the first one's count)
            ctx.shard_sample(0.0, 1e-12, 0.1, ALPHA, z_meas, zero, 0, L.W_LOG)
        got[mode] = ctx.get_particles(False, False, True)[2]
        ctx.close()
    assert np.isfinite(got["0"]).mean() > 0.8
    assert np.array_equal(got["0"], got["1"], equal_nan=True) and np.array_equal(got["0"], got["auto"], equal_nan=True)
    assert np.array_equal(got["0"], got["1+every"], equal_nan=True)


@pytest.mark.parametrize("seed", range(48))
def test_float32_sector_query_on_random_beam_sets(seed):
    _random_beam_set_case(seed, False)


@pytest.mark.parametrize("seed", range(24))
def test_float64_sector_query_on_random_beam_sets(seed):
    """The same beam sets through the float64 instantiation with the raw product (one kernel holds the interval path,
    the parked edge cells and the entry-by-entry redo): weights and log-weights to 1e-11."""
    _random_beam_set_case(seed, True)


def _random_beam_set_case(seed, f64):
    """Random beam sets through the float32 throughput path -- few or many beams (up to 700: past the interval-only
    kernel's fixed layouts), evenly spaced or scattered, with repeated angles, narrow or wide sectors (many intervals per
    sector, or too many for the tables), headings from -14 to 14 rad, odd particle counts: the log-weights against the
    beam model applied to the staged query's ranges of the same poses (mc_localization.py:116-171, :235-240)."""
    from slam_amd import workload as W
    L = _lib()
    rng = np.random.RandomState(100 + seed)
    grid = W.synthetic_map(256)
    kind = seed % 5
    if kind == 0:                                   # evenly spaced, any count
        B = int(rng.choice([1, 2, 7, 90, 200, 400, 700]))
        angles = [k * 360.0 / B for k in range(B)]
    elif kind == 1:                                 # scattered
        B = int(rng.randint(1, 60))
        angles = list(rng.uniform(0.0, 360.0, B))
    elif kind == 2:                                 # scattered with repeats and near-coincident angles
        B = int(rng.randint(2, 40))
        a = rng.uniform(0.0, 360.0, B)
        a[1::3] = a[0::3][:len(a[1::3])]
        a[2::5] = np.minimum(a[0::5][:len(a[2::5])] + 1e-7, 359.9999)
        angles = list(a)
    elif kind == 3:                                 # an arc only, unsorted
        B = int(rng.randint(3, 120))
        angles = list(rng.permutation(np.linspace(10.0, 10.0 + rng.uniform(20, 300), B)))
    else:                                           # evenly spaced, offset from zero
        B = int(rng.choice([13, 45, 180, 360]))
        angles = [(k + 0.37) * 360.0 / B for k in range(B)]
    half = float(rng.choice([0.5, 2.0, 5.0, 5.0, 12.0, 40.0])) * np.pi / 180.0
    M = int(rng.choice([1, 63, 65, 1000, 4097]))
    X = W.uniform_particles(grid, max(M, 8), seed=seed)[:M]
    X[:, 2] = rng.uniform(-14.0, 14.0, M)
    X = X.astype(np.float32).astype(np.float64)
    ctx = L.Context(0, L.F64 if f64 else L.F32)
    ctx.set_map(grid, W.RES, W.RES)
    w4 = np.array([0.97, 0.005, 0.020, 0.005]); w4 /= w4.sum()
    ctx.set_sensor(angles, W.Z_MAX, 0.15, 1.0, w4, half)
    z_meas = ctx.raycast(np.array([3.0, 4.5, 0.3]), want_hits=False)[0][0]
    ctx.set_particles(X)
    ctx.shard_sample(0.0, 1e-12, 0.1, ALPHA, z_meas, np.zeros((M, 3, 12), np.int16), 0, L.W_PRODUCT if f64 else L.W_LOG)
    Xd, wd, logw = ctx.get_particles(True, f64, True)
    R, _ = ctx.raycast(Xd, want_hits=False)
    wo, lwo, P = O.weights_from_ranges(R, z_meas, O.free_cell(Xd[:, 0], Xd[:, 1], grid), O.SensorParams())
    fin = np.isfinite(lwo)
    assert np.array_equal(np.isfinite(logw), fin)
    if f64:
        assert np.allclose(wd, wo, rtol=1e-11, atol=0.0)
        size = np.sum(np.abs(np.log(P[fin])), axis=1) if fin.any() else np.zeros(0)
        assert np.all(np.abs(logw[fin] - lwo[fin]) <= 1e-12 * np.maximum(1.0, size))
        ctx.close()
        return
    if fin.any():
        diff = np.abs(logw[fin] - lwo[fin])
        assert np.all(diff <= 1e-5 * np.maximum(1.0, np.abs(lwo[fin]))), (kind, B, half, M, diff.max())   # 1e-5 of the log-weight
    ctx.close()
