"""Monte Carlo localization (Table 8.2) with the call surface of the reference's
08_grid_and_montecarlo_localization/mc_localization.py, computed on MI355X.

Drop-in use:

    from slam_amd.mc_localization import MCLocalization, res_x, res_y
    mcl = MCLocalization(x0, map_g)          # same constructor (mc_localization.py:36)
    mcl.initialize(x0)
    mcl.sample([1.5, 0], 0.1); mcl.resample()

`sample` (motion sampling + sector range query + beam likelihood for every
particle), the weight normalisation and the low-variance resampling run as HIP
kernels through libmclhip.so.  What stays on the host is what the reference
does once per step for a single pose or draws from numpy's global random
stream: the true-pose update, the random integers / uniform draw that feed the
kernels (so a seeded run consumes `np.random` exactly as the reference does),
and `initialize`'s rejection sampling.
"""
import warnings

import numpy as np

from . import _lib
from .range_sensor import RangeSensor

# module constants, mc_localization.py:9-18
res_x = 0.15
res_y = 0.15
max_x = 15
max_y = 15
max_sensor_range = 5
alpha = [4, 2, 4, 4, 4, 2]

_DEFAULT_BEAMS = [0, 15, 30, 45, 60, 75, 90, 345, 330, 315, 300, 285, 270]  # mc_localization.py:45-48


def _draw_ints(shape):
    """np.random.random_integers(-1000, 1000, ...) without its deprecation noise (same stream)."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        return np.random.random_integers(-1000, 1000, size=shape)


def sample_normal_distribution(b):
    """Table 5.4, approximately normal, zero mean, 'variance' b (mc_localization.py:20-24)."""
    k = _draw_ints(12)
    return b / 6.0 * (k * 0.001).sum()


def sample_triangular_distribution(b):
    """Table 5.4, triangular (mc_localization.py:26-31)."""
    k1 = _draw_ints(None) * 0.001
    k2 = _draw_ints(None) * 0.001
    return b * k1 * k2


sample = sample_normal_distribution


def advance_true_pose(xt_1, ut, dt=0.1):
    """Noise-free velocity model for ONE pose (the v_hat = v, w_hat = w, gamma_hat = 0 case of
    mc_localization.py:74-104).  Host arithmetic on purpose: it keeps the dtype of xt_1's elements,
    as the reference does (its demo's first true pose is float32, mc_localization.py:306)."""
    vt, wt = ut
    x, y, th = xt_1[0], xt_1[1], xt_1[2]
    xt = np.zeros([3])
    if abs(wt) > 0:
        xt[0] = x - vt / wt * np.sin(th) + vt / wt * np.sin(th + wt * dt)
        xt[1] = y + vt / wt * np.cos(th) - vt / wt * np.cos(th + wt * dt)
        xt[2] = th + wt * dt
    else:
        xt[0] = x + vt * np.cos(th) * dt
        xt[1] = y + vt * np.sin(th) * dt
        xt[2] = th
    return xt


class MCLocalization:
    """Particle filter localisation on an occupancy grid.

    Positional arguments and attributes follow mc_localization.py:35-56.  New
    keyword arguments: `M` (number of particles), `sensor_angles` (degrees),
    `dtype` ('f64' | 'f32': precision of the particle state on the device),
    `device`, `log_weights` (normalise exp(logw - max) instead of the raw
    product: same weights, no underflow with hundreds of beams), `noise`
    ('numpy': integers drawn from np.random as the reference does; 'device':
    counter-based generator on the GPU, for large M).
    """

    def __init__(self, x_gt, map, M=500, sensor_angles=None, dtype="f64", device=0, log_weights=False,
                 noise="numpy"):
        self.x_gt = x_gt
        self.M = int(M)
        self.map = map
        self.sensor_angles = list(_DEFAULT_BEAMS if sensor_angles is None else sensor_angles)
        n = len(self.sensor_angles)
        self.sensor_ranges = [0] * n
        self.true_ranges = [0] * n
        self.range_sensor = RangeSensor(device)
        self.range_sensor.z_max = max_sensor_range
        self.log_weights = bool(log_weights)
        self.noise = noise
        if noise not in ("numpy", "device"):
            raise ValueError("noise must be 'numpy' or 'device'")
        self._ctx = _lib.Context(device, _lib.F32 if dtype in ("f32", "float32", np.float32) else _lib.F64)
        self._map_sig = None
        self._sensor_sig = None
        self._obstacles = None
        self._Xh = np.zeros([self.M, 3])
        self._wh = np.ones(self.M) / self.M
        self._host_fresh = True     # host copies hold the live values
        self._dev_fresh = False     # device buffers hold the live values
        self.last_idx = None
        self._push_config()

    # -- state mirrors -----------------------------------------------------
    def _pull(self):
        if not self._host_fresh:
            self._Xh, self._wh, _ = self._ctx.get_particles()
            self._host_fresh = True

    @property
    def X(self):
        """Particles [M,3].  Reading hands out the host copy, which the caller may edit in place
        (initialize does), so the device copy is refreshed from it before the next kernel."""
        self._pull()
        self._dev_fresh = False
        return self._Xh

    @X.setter
    def X(self, value):
        self._pull()
        self._Xh = np.array(value, dtype=np.float64).reshape(-1, 3)
        self.M = len(self._Xh)
        if len(self._wh) != self.M:
            self._wh = np.ones(self.M) / self.M
        self._dev_fresh = False

    @property
    def w(self):
        self._pull()
        self._dev_fresh = False
        return self._wh

    @w.setter
    def w(self, value):
        self._pull()
        self._wh = np.array(value, dtype=np.float64).reshape(-1)
        self._dev_fresh = False

    @property
    def log_w(self):
        """sum_k log p_k of the last sample(), -inf where the motion left the free space."""
        return self._ctx.get_particles(False, False, True)[2]

    @property
    def obstacles(self):
        """float32 [N,2] centres of the cells equal to 0, row-major (mc_localization.py:61-71).
        Informational: the kernels read the grid itself."""
        if self._obstacles is None:
            rows, cols = np.nonzero(np.asarray(self.map) == 0)
            lut = np.stack([cols, rows], 1).astype(np.float32)
            if len(lut):
                lut[:, 0] *= res_x
                lut[:, 1] *= res_y
            self._obstacles = lut
        return self._obstacles

    def reset_obstacle_lut(self):
        self._obstacles = None
        self._map_sig = None

    def make_obstacle_lut(self, map):
        self.map = map
        self.reset_obstacle_lut()
        self._push_config()

    # -- device configuration --------------------------------------------
    def _push_config(self):
        g = np.asarray(self.map)
        sig = (g.shape, g.ctypes.data if g.flags["C_CONTIGUOUS"] else None, float(res_x), float(res_y))
        if self._map_sig is None or sig != self._map_sig:
            self._ctx.set_map(g, res_x, res_y)
            self._map_sig = sig
        z_max, sigma, lam, w4 = self.range_sensor.constants()
        ssig = (tuple(float(a) for a in self.sensor_angles), z_max, sigma, lam, w4)
        if ssig != self._sensor_sig:
            self._ctx.set_sensor(self.sensor_angles, z_max, sigma, lam, w4, 5 * np.pi / 180.0)
            self._sensor_sig = ssig

    def _push_particles(self):
        if not self._dev_fresh:
            self._ctx.set_particles(self._Xh, self._wh)
            self._dev_fresh = True

    # -- reference surface ---------------------------------------------------
    def sample_motion_model_velocity(self, xt_1, ut, noise=False, dt=0.1):
        """Table 5.3 for one pose -> (pose, 1.0 on a free cell else 0.0)  (mc_localization.py:74-114)."""
        vt, wt = ut
        if noise:
            self._push_config()
            ints = _draw_ints((1, 3, 12))
            out, pm = self._ctx.motion(np.asarray(xt_1, np.float64).reshape(1, 3), vt, wt, dt, alpha, ints)
            return out[0], float(pm[0])
        xt = advance_true_pose(xt_1, ut, dt)
        g = np.asarray(self.map)
        u = int(xt[0] / res_x + 0.5)
        v = int(xt[1] / res_y + 0.5)
        free = 0 <= u < g.shape[1] and 0 <= v < g.shape[0] and g[v, u] == 255
        return xt, (1.0 if free else 0.0)

    def measure(self, xt, noise=False):
        """Ranges of all beams from pose xt (mc_localization.py:116-171): distance to the nearest
        obstacle-cell centre inside each beam's +-5 degree sector, z_max if none."""
        self._push_config()
        # one pose per filter step: the variant that allocates no device memory per call
        R, _ = self._ctx.raycast(np.asarray(xt, np.float64).reshape(1, 3), want_hits=False, static=True)
        out = [np.float64(r) for r in R[0]]
        if noise:
            out = [r + np.random.randn() * 0.15 for r in out]
        return out

    def measure_batch(self, poses):
        """(ranges [n,B], hit cells [n,B]) for many poses at once (new; stage entry mclhip_raycast)."""
        self._push_config()
        return self._ctx.raycast(poses, want_hits=True)

    def initialize(self, x0=None, on_device=False, seed=0):
        """Rejection-sample M poses on free cells (mc_localization.py:173-219); consumes
        np.random exactly as the reference does.  on_device=True draws the cloud on the GPU instead
        (same distribution, the device's own generator): for particle counts at which a Python loop
        per particle is not an option."""
        if on_device:
            self._push_config()
            self._ctx.init_particles(self.M, x0, max_x, max_y, seed)
            self._host_fresh, self._dev_fresh = False, True
            return
        g = np.asarray(self.map)
        H, W = g.shape
        X = np.zeros([self.M, 3])
        for m in range(self.M):
            while True:
                if x0 is None:
                    x = np.random.rand() * max_x
                    y = np.random.rand() * max_y
                else:
                    x = np.random.randn() * max_x * 0.1 + x0[0]
                    y = np.random.randn() * max_y * 0.1 + x0[1]
                u = int(x / res_x + 0.5)
                v = int(y / res_y + 0.5)
                if x0 is None:
                    # the uniform branch only guards the upper bounds (mc_localization.py:184)
                    if not (u >= W or v >= H or g[v, u] < 255):
                        break
                elif 0 <= u < W and 0 <= v < H and g[v, u] == 255:
                    break
            if x0 is None:
                a = np.random.rand() * np.pi * 2.0
            else:
                a = np.random.randn() * np.pi + x0[2]
            X[m] = (x, y, a)
        self.X = X
        self.w = np.ones(self.M) / self.M

    def sample(self, ut, dt=0.1, noise_ints=None):
        """One prediction + measurement update (mc_localization.py:221-240).  `noise_ints`
        ([M,3,12], optional) overrides the integers drawn for the motion noise."""
        self._push_config()
        self.x_gt, _ = self.sample_motion_model_velocity(self.x_gt, ut, noise=False, dt=dt)
        self.sensor_ranges = self.measure(self.x_gt)
        self._push_particles()
        vt, wt = ut
        seed = 0
        if noise_ints is None:
            if self.noise == "numpy":
                noise_ints = _draw_ints((self.M, 3, 12))
            else:
                seed = int(np.random.randint(0, 2 ** 31 - 1)) * 2654435761 + 1
        self._ctx.sample(vt, wt, dt, alpha, self.sensor_ranges, noise_ints, seed)
        self._host_fresh = False
        self._dev_fresh = True

    def resample(self, r=None):
        """Normalise the weights and draw M particles with the low-variance sampler
        (mc_localization.py:242-262).  `r` (optional) overrides the uniform draw."""
        self._push_particles()
        if r is None:
            r = np.random.rand() * 1 / self.M
        mode = _lib.W_LOG if self.log_weights else _lib.W_PRODUCT
        self.w_sum = self._ctx.normalise(mode)
        self.last_idx = self._ctx.resample(r)
        self._host_fresh = False
        self._dev_fresh = True

    # -- plotting (mc_localization.py:264-301) ----------------------------
    def plot(self):
        import matplotlib.pyplot as plt
        self.draw_robot(self.x_gt, color="g", sensor_readings=True)
        X = self.X
        px, py = X[:, 0] / res_x, X[:, 1] / res_y
        plt.scatter(px, py, c="b")
        plt.plot([px, px + 0.5 * np.cos(X[:, 2]) / res_x], [py, py + 0.5 * np.sin(X[:, 2]) / res_y], c="b")

    def draw_robot(self, xt, color="g", sensor_readings=False):
        import matplotlib.pyplot as plt
        x, y, th = xt[0], xt[1], xt[2]
        plt.scatter(x / res_x, y / res_y, c=color)
        plt.plot([x / res_x, (x + np.cos(th)) / res_x], [y / res_y, (y + np.sin(th)) / res_y], c=color)
        if sensor_readings:
            for deg, rng in zip(self.sensor_angles, self.sensor_ranges):
                b = deg * np.pi / 180.0 + th
                plt.plot([x / res_x, (x + rng * np.cos(b)) / res_x], [y / res_y, (y + rng * np.sin(b)) / res_y], c="r")
