"""Velocity motion model of chapter 5 with the call surface of the reference's
05_robot_motion/sample_motion_model_velocity.py (functions and module constants; the script's
plotting demo is not part of it).

`sample_motion_model_velocity` is Table 5.3 as printed -- the arc update for every w_hat, no
straight-line branch (a w_hat of exactly zero gives inf / NaN, as numpy's float64 division does
in the reference) and no map.  It runs on the GPU through `mclhip_motion_book`; the host draws the
3 x 12 integers from np.random exactly as the reference's `sample` does, so a seeded run consumes
the stream identically.  `sample_motion_model_velocity_batch` moves many poses in one call.

The densities (`prob_*`, `motion_model_velocity`, Tables 5.1 / 5.2) are one-pose helpers of the
reference's demo; they stay host arithmetic (the histogram filter evaluates its own variant of the
density on the GPU, see slam_amd/robot.py).

There is no CPU fallback for the sampler: without the library or a GPU it raises.
"""
import numpy as np

from . import _lib
from .mc_localization import _draw_ints, sample_normal_distribution, sample_triangular_distribution  # noqa: F401

delta_t = 1 / 10                                  # sample_motion_model_velocity.py:8
alpha = [0.2, 0.1, 0.2, 0.2, 0.1, 0.1]            # :11

sample = sample_normal_distribution               # :88

_ctx = None


def _context(device=0):
    global _ctx
    if _ctx is None:
        _ctx = _lib.Context(device, _lib.F64)
    return _ctx


def prob_normal_distribution(a, b):
    """Table 5.2: zero-centred normal density with variance b (:13-16)."""
    return 1 / np.sqrt(2 * np.pi * b) * np.exp(-1 / 2 * (a ** 2) / b)


def prob_triangular_distribution(a, b):
    """Table 5.2: triangular density with variance b (:18-24)."""
    p = (np.sqrt(6 * b) - abs(a)) / (6 * b)
    p *= abs(a) <= np.sqrt(6 * b)
    return p


prob = prob_normal_distribution                   # :38


def motion_model_velocity(xt, ut, xt_1, dt=1 / 10):
    """Table 5.1, p(xt | ut, xt_1) for one pose pair (:40-71).  Host helper."""
    x1, y1, ha1 = xt
    x0, y0, ha0 = xt_1
    num = (x0 - x1) * np.cos(ha0) + (y0 - y1) * np.sin(ha0)
    den = (y0 - y1) * np.cos(ha0) - (x0 - x1) * np.sin(ha0)
    mu = 1 / 2 * num / den
    cx = (x0 + x1) / 2 + mu * (y0 - y1)
    cy = (y0 + y1) / 2 + mu * (x1 - x0)
    cr = np.sqrt((x0 - cx) ** 2 + (y0 - cy) ** 2)
    d_ha = np.arctan2(y1 - cy, x1 - cx) - np.arctan2(y0 - cy, x0 - cx)
    v_hat = d_ha / dt * cr
    w_hat = d_ha / dt
    g_hat = (ha1 - ha0) / dt - w_hat
    v, w = ut
    pv = prob(v - v_hat, alpha[0] * abs(v) + alpha[1] * abs(w))
    pw = prob(w - w_hat, alpha[2] * abs(v) + alpha[3] * abs(w))
    pg = prob(g_hat, alpha[4] * abs(v) + alpha[5] * abs(w))
    return pv * pw * pg


def sample_motion_model_velocity_batch(ut, X, dt=1 / 10, noise_ints=None, device=0):
    """Table 5.3 for every row of X [n,3] in one kernel launch.  noise_ints [n,3,12]: the integers
    of the three `sample` draws of every pose; None draws them from np.random in the order n
    successive calls of the reference's function would."""
    X = np.ascontiguousarray(X, np.float64).reshape(-1, 3)
    if noise_ints is None:
        noise_ints = _draw_ints((len(X), 3, 12))
    v, w = ut
    return _context(device).motion_book(X, float(v), float(w), float(dt), alpha, noise_ints)


def sample_motion_model_velocity(ut, xt_1, dt=1 / 10):
    """Table 5.3: one pose xt sampled from xt_1 and the control ut = (v, w) (:90-108)."""
    return sample_motion_model_velocity_batch(ut, np.asarray(xt_1, np.float64).reshape(1, 3), dt)[0]
