"""The chapter-4 low-variance sampler and the chapter-5 motion sampler (the stand-alone scripts the
particle filter's stages come from) through the C ABI and through their drop-in modules, against the
golden vectors captured from the reference's own functions.  Needs an MI355X: `-m gpu`.

Bars: drawn particles, carried weights and the position histogram bit-exact; sampled poses within
1e-12 relative (device sin/cos differ from numpy's by ulps), NaN / inf pattern identical.
"""
import numpy as np
import pytest

from util import load, rel_err

pytestmark = pytest.mark.gpu


def test_motion_sampler_of_chapter5():
    from slam_amd import _lib as L
    g = load("motion05")
    ctx = L.Context(0, L.F64)                       # no map, no sensor: the stage needs neither
    dt = float(g["dt"])
    for c, ut in enumerate(g["controls"]):
        X = ctx.motion_book(g["X"], ut[0], ut[1], dt, g["alpha"], g["noise"][c])
        assert rel_err(X, g["X_out"][c]) < 1e-12
    Z = ctx.motion_book(g["X"], 1.5, 0.0, dt, g["alpha"], np.zeros((len(g["X"]), 3, 12), np.int16))
    assert np.array_equal(np.isnan(Z), np.isnan(g["X_out_zero"]))
    assert rel_err(Z[:, 2], g["X_out_zero"][:, 2]) < 1e-12
    # noise-free call; a single pose; a large batch
    X0 = ctx.motion_book(g["X"][:1], 10.0, 2.0, dt, g["alpha"], None)
    th = g["X"][0, 2]
    assert rel_err(X0[0], [g["X"][0, 0] - 5 * np.sin(th) + 5 * np.sin(th + 2.0 * dt),
                           g["X"][0, 1] + 5 * np.cos(th) - 5 * np.cos(th + 2.0 * dt), th + 2.0 * dt]) < 1e-12
    big = np.tile(g["X"], (300, 1))
    Xb = ctx.motion_book(big, 1.5, 0.5, dt, g["alpha"], np.tile(g["noise"][1], (300, 1, 1)))
    assert np.array_equal(Xb[:len(g["X"])], Xb[-len(g["X"]):]) and rel_err(Xb[:len(g["X"])], g["X_out"][1]) < 1e-12
    ctx.close()


def test_motion_sampler_dropin_consumes_the_stream_like_the_reference():
    from slam_amd import sample_motion_model_velocity as mv
    g = load("motion05")
    ut = g["controls"][0]
    np.random.seed(98)                              # the generating run's seed (tools/gen_golden.py)
    for m in range(40):
        xt = mv.sample_motion_model_velocity(ut, g["X"][m], dt=mv.delta_t)
        assert xt.shape == (3,) and rel_err(xt, g["X_out"][0, m]) < 1e-12
    np.random.seed(98)
    Xb = mv.sample_motion_model_velocity_batch(ut, g["X"], dt=mv.delta_t)
    assert rel_err(Xb, g["X_out"][0]) < 1e-12


def test_low_variance_sampler_of_chapter4():
    from slam_amd import particle_filter as pf
    g = load("particle_filter04")
    seeds = [41, 42, 43, 44, 45, 46]
    for n in range(int(g["n_cases"])):
        w = g["c%d_w" % n]
        flt = pf.ParticleFilter(len(w), np.ones(int(g["x_max"])) / int(g["x_max"]))
        flt.X = list(g["c%d_X" % n])
        flt.w = w.copy()
        np.random.seed(seeds[n])
        flt.resample()
        assert np.array_equal(np.array(flt.X), g["c%d_X_out" % n]), n
        assert np.array_equal(np.array(flt.w), g["c%d_w_out" % n]), n
        assert np.array_equal(flt.px, g["c%d_px" % n]), n
        assert np.all(np.diff(flt.last_idx) >= 0)


def test_particle_filter_dropin_follows_the_reference_loop():
    """Three iterations of the script's loop body (particle_filter.py:110-162, plotting aside)."""
    from slam_amd import particle_filter as pf
    g = load("particle_filter04")
    M = 50
    np.random.seed(7)
    flt = pf.ParticleFilter(M, np.ones(pf.x_max) / pf.x_max)
    for it in range(3):
        if it == 0:
            flt.sample_uniform()
        else:
            flt.sample_gaussian()
        pz = np.zeros([pf.x_max])
        for m in range(M):
            pz[int(np.clip(flt.X[m], a_min=0, a_max=pf.x_max - 1) + 0.5)] += 1
        pz /= M
        for m in range(M):
            z = int(np.clip(flt.X[m], a_min=0, a_max=pf.x_max - 1) + 0.5)
            flt.w[m] = pf.normal_dist(pf.x_obj, 1, z) * pz[z] / (flt.px[z] + 1e-6)
        assert np.array_equal(np.array(flt.X, np.float64), g["loop%d_X_in" % it])
        assert np.array_equal(np.array(flt.w, np.float64), g["loop%d_w_in" % it])
        flt.resample()
        assert np.array_equal(np.array(flt.X), g["loop%d_X" % it])
        assert np.array_equal(np.array(flt.w), g["loop%d_w" % it])
        assert np.array_equal(flt.px, g["loop%d_px" % it])
