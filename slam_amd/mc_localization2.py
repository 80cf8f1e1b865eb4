"""Augmented Monte Carlo localization (Table 8.3) with the call surface of the reference's
08_grid_and_montecarlo_localization/mc_localization2.py, computed on MI355X.

Same drop-in rules as `slam_amd.mc_localization`; what this variant adds to the plain filter:
noisy predicted ranges inside the weight (`measure(X[m], noise=True)`, mc_localization2.py:237),
the running weight means `w_avg`, `w_slow`, `w_fast` (:246-249) and random-pose injection with
probability max(0, 1 - w_fast/w_slow) inside `resample` (:262-296).  All random draws come
from numpy's global stream in the order the reference consumes them, so they are made on the
host, one particle at a time, and handed to the kernels as arrays.  `noise="device"` draws them
on the GPU instead (motion integers, one normal per particle and beam, the injection draws): the
same distributions from the device's own generator, for particle counts at which a Python loop
per particle and beam is not an option.
"""
import numpy as np

from . import _lib
from . import mc_localization as _base
from .mc_localization import (advance_true_pose, alpha, max_sensor_range, max_x, max_y, res_x,  # noqa: F401
                              res_y)


def sample_normal_distribution(b):
    """Table 5.4 with np.random.randint(-1000, 1000, 12): the upper bound is exclusive here
    (mc_localization2.py:20-24)."""
    k = np.random.randint(-1000, 1000, size=12)
    return b / 6.0 * (k * 0.001).sum()


def sample_triangular_distribution(b):
    k1 = np.random.randint(-1000, 1000) * 0.001
    k2 = np.random.randint(-1000, 1000) * 0.001
    return b * k1 * k2


sample = sample_normal_distribution


class MCLocalization(_base.MCLocalization):
    """Attributes added by mc_localization2.py:42-46: w_avg, w_slow, w_fast, a_slow, a_fast."""

    def __init__(self, x_gt, map, **kw):
        super().__init__(x_gt, map, **kw)
        self.w_avg = 0
        self.w_slow = 0
        self.w_fast = 0
        self.a_slow = 0.001
        self.a_fast = 0.9
        self.p_random = 0.0

    def sample_motion_model_velocity(self, xt_1, ut, noise=False, dt=0.1):
        if not noise:
            return super().sample_motion_model_velocity(xt_1, ut, noise=False, dt=dt)
        self._push_config()
        ints = np.stack([np.random.randint(-1000, 1000, size=12) for _ in range(3)])[None]
        out, pm = self._ctx.motion(np.asarray(xt_1, np.float64).reshape(1, 3), ut[0], ut[1], dt, alpha, ints)
        return out[0], float(pm[0])

    def sample(self, ut, dt=0.1, noise_ints=None, range_noise=None):
        """mc_localization2.py:224-249.  Per particle the reference draws 3 x 12 integers for the
        motion and then one normal deviate per beam for the predicted ranges; the same draws, in
        the same order, are made here unless both arrays are passed in."""
        self._push_config()
        self.x_gt, _ = super().sample_motion_model_velocity(self.x_gt, ut, noise=False, dt=dt)
        self.sensor_ranges = self.measure(self.x_gt)
        self._push_particles()
        B = len(self.sensor_angles)
        if self.noise == "device" and noise_ints is None and range_noise is None:
            self._seed = int(np.random.randint(0, 2 ** 31 - 1)) * 2654435761 + 1
            self.w_avg = np.float64(self._ctx.sample_augmented_device(ut[0], ut[1], dt, alpha, self.sensor_ranges,
                                                                      self._seed, 0.15))
            self.w_slow = self.w_slow + self.a_slow * (self.w_avg - self.w_slow)
            self.w_fast = self.w_fast + self.a_fast * (self.w_avg - self.w_fast)
            self._host_fresh = False
            self._dev_fresh = True
            return
        if noise_ints is None or range_noise is None:
            noise_ints = np.zeros((self.M, 3, 12), np.int16)
            range_noise = np.zeros((self.M, B))
            for m in range(self.M):
                for j in range(3):
                    noise_ints[m, j] = np.random.randint(-1000, 1000, size=12)
                for k in range(B):
                    range_noise[m, k] = np.random.randn() * 0.15
        self.w_avg = np.float64(self._ctx.sample_augmented(ut[0], ut[1], dt, alpha, self.sensor_ranges, noise_ints,
                                                           range_noise))
        self.w_slow = self.w_slow + self.a_slow * (self.w_avg - self.w_slow)
        self.w_fast = self.w_fast + self.a_fast * (self.w_avg - self.w_fast)
        self._host_fresh = False
        self._dev_fresh = True

    def draw_injections(self):
        """The per-position draws of mc_localization2.py:266-283: one uniform per position, and for
        an injected position a rejection-sampled free cell and a heading."""
        g = np.asarray(self.map)
        H, W = g.shape
        rows, poses = [], []
        for m in range(self.M):
            if np.random.rand() <= self.p_random:
                while True:
                    x = np.random.rand() * max_x
                    y = np.random.rand() * max_y
                    u = int(x / res_x + 0.5)
                    v = int(y / res_y + 0.5)
                    if not (u >= W or v >= H or g[v, u] < 255):
                        break
                rows.append(m)
                poses.append((x, y, np.random.rand() * np.pi * 2.0))
        return np.array(rows, np.int64), np.array(poses, np.float64).reshape(-1, 3)

    def resample(self, r=None, injections=None):
        """mc_localization2.py:254-299."""
        self._push_particles()
        if r is None:
            r = np.random.rand() * 1 / self.M
        self.w_sum = self._ctx.normalise(_lib.W_PRODUCT)
        with np.errstate(divide="ignore", invalid="ignore"):
            self.p_random = max(0.0, 1 - np.float64(self.w_fast) / np.float64(self.w_slow))
        if self.noise == "device" and injections is None:
            seed = int(np.random.randint(0, 2 ** 31 - 1)) * 2654435761 + 7
            self.n_injected = self._ctx.resample_augmented_device(r, self.p_random, max_x, max_y, seed)
            self.last_idx, self.last_injected = None, None
            self._host_fresh = False
            self._dev_fresh = True
            return
        rows, poses = self.draw_injections() if injections is None else injections
        self.last_idx = self._ctx.resample_augmented(r, rows, poses)
        self.last_injected = rows
        self._host_fresh = False
        self._dev_fresh = True
