"""The oracle against the golden vectors captured from the reference
(tools/gen_golden.py).  Bit-exact unless a tolerance is written in the test."""
import numpy as np
import pytest

from oracle import mcl_oracle as O
from util import load, steps

DEMO = ["demo_seed0", "demo_seed1", "demo_seed2"]
SYNTH = ["synth128_b13", "synth256_b180", "synth256_b360"]


def _grid(name, g):
    if name.startswith("demo"):
        return np.load(__import__("os").path.join(__import__("util").GOLDEN, "grid_map_g.npy"))
    return g["map"]


@pytest.mark.parametrize("name", DEMO + SYNTH)
def test_stages_bit_exact(name):
    g = load(name)
    grid = _grid(name, g)
    angles = list(g["sensor_angles"])
    if name.startswith("demo"):
        angles = [int(a) for a in angles]  # the reference list holds Python ints
    sp = O.SensorParams()
    lut, cell = O.obstacle_lut(grid)
    for s in steps(g):
        ut = (float(s["ut"][0]), float(s["ut"][1]))
        if name.startswith("demo"):
            ut = (ut[0], int(ut[1])) if ut[1] == int(ut[1]) else ut
        dt = float(s["dt"])
        # true pose: keeps float32 arithmetic on the demo's first step
        x_in = s["x_gt_in"].astype(np.float32) if bool(s["x_gt_in_is_f32"]) else s["x_gt_in"]
        x_gt, _ = O.motion_single(x_in, ut, dt, grid)
        assert np.array_equal(x_gt, s["x_gt_out"])
        z, hg = O.sector_query(x_gt, lut, cell, angles)
        assert np.array_equal(z, s["z_meas"])
        assert np.array_equal(hg, s["hits_gt"])
        X, pm = O.motion(s["X_in"], ut, dt, grid, s["noise"])
        assert np.array_equal(X, s["X_motion"])
        assert np.array_equal(pm, s["pm"])
        R, Hc = O.sector_query_batch(X, lut, cell, angles)
        assert np.array_equal(R, s["ranges"])
        assert np.array_equal(Hc, s["hits"])
        w, logw, P = O.weights_from_ranges(R, z, pm, sp)
        assert np.array_equal(P, s["p_beam"])
        assert np.array_equal(w, s["w_raw"])
        assert O.numpy_order_sum(w) == s["w_sum"] or (np.isnan(s["w_sum"]) and np.isnan(O.numpy_order_sum(w)))
        if bool(s["resample_ok"]):
            wn = O.normalise(w)
            idx = O.resample_indices(wn, float(s["r"]))
            assert np.array_equal(idx, s["idx"])
            assert np.array_equal(X[idx], s["X_out"])
            assert np.array_equal(wn[idx], s["w_out"])


def test_compute_px_lattice():
    g = load("compute_px")
    sp = O.SensorParams(z_max=float(g["z_max"]))
    assert np.array_equal(np.array([sp.w_hit, sp.w_short, sp.w_max, sp.w_rand]), g["w4"])
    p = O.beam_px(g["z"], g["z_true"], sp)
    assert np.array_equal(p, g["p"])


def test_resample_only_cases():
    g = load("resample_only")
    for c in range(int(g["n_cases"])):
        w = g["c%d_w" % c]
        wn = O.normalise(w)
        idx = O.resample_indices(wn, float(g["c%d_r" % c]))
        assert np.array_equal(idx, g["c%d_idx" % c])
        assert np.array_equal(wn[idx], g["c%d_w_out" % c])
        if len(w) <= 1000:
            assert np.array_equal(O.resample_indices_loop(wn, float(g["c%d_r" % c])), idx)


def test_motion_only():
    g = load("motion_only")
    grid = np.load(__import__("os").path.join(__import__("util").GOLDEN, "grid_map_g.npy"))
    for c, ut in enumerate(g["controls"]):
        X, pm = O.motion(g["X"], (float(ut[0]), float(ut[1])), 0.1, grid, g["noise"][c], alpha=tuple(g["alpha"]))
        assert np.array_equal(X, g["X_out"][c])
        assert np.array_equal(pm, g["pm"][c])
    Z, pz = O.motion(g["X"], (1.5, 0.0), 0.1, grid, np.zeros((len(g["X"]), 3, 12), np.int16), alpha=tuple(g["alpha"]))
    assert np.array_equal(Z, g["X_out_zero_noise"])
    assert np.array_equal(pz, g["pm_zero_noise"])


def test_test_robot_known_answers():
    """The reference's own range-sensing script: every beam it checks reads < 4 m
    there; the float64 oracle must agree on all of them.  The script runs its
    query in float32 (its pose is float32), so ranges agree to 1e-5 only."""
    g = load("test_robot")
    angles = [int(a) for a in g["sensor_angles"]]
    for ci, beam, ok in g["checks"]:
        grid = g["maps"][ci]
        # Robot.make_obstacle_lut thresholds at < 200 (robot.py:199); these maps hold only 0/255
        lut, cell = O.obstacle_lut(grid)
        z, _ = O.sector_query(g["poses"][ci], lut, cell, angles)
        assert (z[beam] < 4) == bool(ok)
        assert np.allclose(z, g["ranges_f32path"][ci], rtol=1e-5, atol=1e-5)


def test_numpy_order_sum_matches_numpy():
    rng = np.random.RandomState(3)
    for n in [1, 7, 8, 12, 129, 500, 8191, 8192, 8193, 20000, 70001]:
        a = rng.rand(n) * rng.rand(n) ** 6
        assert O.numpy_order_sum(a) == np.sum(a)


def test_sum12_matches_numpy_sum():
    rng = np.random.RandomState(4)
    ints = rng.randint(-1000, 1001, size=(500, 12))
    s = O.sum12(ints)
    for i in range(len(ints)):
        assert s[i] == (ints[i] * 0.001).sum()


@pytest.mark.parametrize("name", ["amcl_seed0", "amcl_seed4"])
def test_augmented_mcl_stages_bit_exact(name):
    """mc_localization2.py: noisy predicted ranges, running weight means, injection."""
    g = load(name)
    grid = np.load(__import__("os").path.join(__import__("util").GOLDEN, "grid_map_g.npy"))
    angles = [int(a) for a in g["sensor_angles"]]
    sp = O.SensorParams()
    lut, cell = O.obstacle_lut(grid)
    for s in steps(g):
        x_in = s["x_gt_in"].astype(np.float32) if bool(s["x_gt_in_is_f32"]) else s["x_gt_in"]
        x_gt, _ = O.motion_single(x_in, (1.5, 0), 0.1, grid)
        assert np.array_equal(x_gt, s["x_gt_out"])
        z, _ = O.sector_query(x_gt, lut, cell, angles)
        assert np.array_equal(z, s["z_meas"])
        X, pm = O.motion(s["X_in"], (1.5, 0), 0.1, grid, s["noise"])
        assert np.array_equal(X, s["X_motion"])
        R, _ = O.sector_query_batch(X, lut, cell, angles)
        w, _ = O.amcl_weights(R, s["range_noise"], z, pm, sp)
        assert np.array_equal(w, s["w_raw"])
        w_avg = O.sequential_mean(w)
        assert w_avg == s["w_avg"]
        ws, wf = O.amcl_rates(s["w_slow_in"], s["w_fast_in"], w_avg)
        assert ws == s["w_slow"] and wf == s["w_fast"]
        assert max(0.0, 1 - s["w_fast_rs"] / s["w_slow_rs"]) == s["p_random"]
        wn = O.normalise(w)
        Xo, wo, _ = O.amcl_resample(X, wn, float(s["r"]), s["inj_rows"], s["inj_poses"])
        assert np.array_equal(Xo, s["X_out"])
        assert np.array_equal(wo, s["w_out"])


def test_chapter5_motion_sampler_bit_exact():
    """Table 5.3 without the straight-line branch (05_robot_motion/sample_motion_model_velocity.py:90-108)."""
    g = load("motion05")
    al = tuple(g["alpha"])
    for c, ut in enumerate(g["controls"]):
        assert np.array_equal(O.motion_book(g["X"], ut, float(g["dt"]), g["noise"][c], al), g["X_out"][c])
    Z = O.motion_book(g["X"], (1.5, 0.0), float(g["dt"]), np.zeros((len(g["X"]), 3, 12), np.int16), al)
    assert np.array_equal(Z, g["X_out_zero"], equal_nan=True)
    assert np.all(np.isnan(Z[:, :2])) and np.all(np.isfinite(Z[:, 2]))   # 1.5 / 0 -> inf - inf


def test_chapter4_low_variance_sampler_bit_exact():
    """ParticleFilter.resample of 04_nonparametric_filters/particle_filter.py:58-81."""
    g = load("particle_filter04")
    assert int(g["n_cases"]) == 6
    for n in range(int(g["n_cases"])):
        w = g["c%d_w" % n]
        wn = w / np.sum(w)
        idx = O.resample_indices(wn, float(g["c%d_r" % n]))
        assert np.array_equal(g["c%d_X" % n][idx], g["c%d_X_out" % n])
        assert np.array_equal(wn[idx], g["c%d_w_out" % n])


def test_chapter5_density_helper_matches_reference():
    """Host helper of the drop-in module (Table 5.1); pure numpy, needs no GPU."""
    from slam_amd import sample_motion_model_velocity as mv
    g = load("motion05")
    ut = np.array([10.0, np.pi / 4 * 10])
    d = np.array([mv.motion_model_velocity(g["dens_xt"][i], ut, g["dens_xt1"][i], dt=float(g["dt"]))
                  for i in range(len(g["dens"]))])
    assert np.array_equal(d, g["dens"])
    assert mv.alpha == list(g["alpha"]) and mv.delta_t == float(g["dt"])
