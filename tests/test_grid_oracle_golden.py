"""The grid-localisation oracle against the golden vectors captured from the reference's Robot
(tools/gen_golden.py `grid`).  Everything is bit-exact: the initial belief, the true-pose update,
the float32 sensing path with the tabulated-pdf draw, predict() and update()."""
import numpy as np
import pytest

from oracle import grid_oracle as G
from util import load, steps

CASES = ["grid_seed0", "grid_seed1"]


def _ut(s):
    v, w = float(s["ut"][0]), float(s["ut"][1])
    return (v, int(w)) if w == int(w) else (v, w)   # the driver passes wt = 0 as a Python int


@pytest.mark.parametrize("name", CASES)
def test_initial_belief_and_true_pose(name):
    g = load(name)
    assert np.array_equal(G.initial_belief(g["map"], 4), g["pk0"])
    for s in steps(g):
        x = s["x_gt_in"].copy()
        assert x.dtype == np.float32
        G.advance_true_pose(x, _ut(s))
        assert np.array_equal(x, s["x_gt_out"])


@pytest.mark.parametrize("name", CASES)
def test_sensing_at_the_true_pose(name):
    """float32 geometry (robot.py:226-275 with a float32 pose) and the draw from the tabulated
    beam model (range_sensor.py:69-126) given the recorded uniform numbers."""
    g = load(name)
    lut, _ = G.obstacle_lut(g["map"])
    for s in steps(g):
        z_meas, z_true = G.sense(s["x_gt_out"], lut, s["uniforms"])
        assert np.array_equal(np.array([np.float64(v) for v in z_true]), s["z_true"])
        assert np.array_equal(np.array(z_meas), s["z_meas"])


@pytest.mark.parametrize("name", CASES)
def test_predict_bit_exact(name):
    g = load(name)
    for s in steps(g):
        assert np.array_equal(G.predict(s["pk_in"], g["map"], _ut(s), 4), s["pk_pred"])


def test_update_bit_exact():
    g = load("grid_seed0")
    s = steps(g)[1]
    pk, total, _ = G.update(s["pk_pred"], g["map"], [float(v) for v in s["z_meas"]], 4)
    assert pk.dtype == np.float32
    assert np.array_equal(pk, s["pk_out"])
    assert np.float32(total) == np.float32(np.sum(_unnormalised(s, g)))


def _unnormalised(s, g):
    return G.update(s["pk_pred"], g["map"], [float(v) for v in s["z_meas"]], 4)[2]


def test_float32_sum_restatement_equals_numpy():
    rng = np.random.RandomState(5)
    for n in (1, 7, 8, 129, 8191, 8192, 8193, 40000, 100003):
        a = (rng.rand(n) ** 8).astype(np.float32)
        assert G.f32_sum(a) == np.sum(a)
    a = (rng.rand(30, 31, 4) ** 4).astype(np.float32)
    assert G.f32_sum(a) == a.sum()


def test_per_cell_ranges_equal_the_literal_per_pose_form():
    g = load("grid_seed0")
    grid = g["map"][40:70, 30:60].copy()
    lut, _ = G.obstacle_lut(grid)
    ii, jj, Z = G.predicted_ranges(grid, 4)
    for n in range(0, len(ii), 17):
        for d in range(4):
            xt = np.array([[jj[n] * G.RES_X, ii[n] * G.RES_Y, G.heading(d)]]).T
            assert np.array_equal(np.array(G.ranges_at(xt, lut), np.float64), Z[n, d])


def test_translation_invariant_coefficients_are_close_enough():
    """The GPU evaluates the motion density once per (offset, heading pair) instead of per cell.
    Against the per-cell evaluation the difference is rounding of absolute coordinates only:
    below 1e-11 relative on the reference's map."""
    g = load("grid_seed0")
    s = steps(g)[1]
    D = 4
    oi, oj, od = np.meshgrid(np.arange(-3, 4), np.arange(-3, 4), np.arange(D), indexing="ij")
    ha0 = np.vectorize(G.heading)(od)
    pk = s["pk_in"].astype(np.float64)
    H, W = g["map"].shape
    out = np.zeros((H, W, D))
    for d in range(D):
        K = G.motion_density(0.0, 0.0, G.heading(d), _ut(s), oj * G.RES_X, oi * G.RES_Y, ha0)
        acc = np.zeros((H - 6, W - 6))
        for a in range(7):
            for b in range(7):
                for nd in range(D):
                    acc += K[a, b, nd] * pk[a:a + H - 6, b:b + W - 6, nd]
        out[3:H - 3, 3:W - 3, d] = acc * (g["map"][3:H - 3, 3:W - 3] >= 200)
    ref = s["pk_pred"]
    nz = ref != 0
    assert np.array_equal(out != 0, nz)
    assert np.max(np.abs(out - ref)[nz] / ref[nz]) < 1e-11
