"""Grid (histogram) localisation with the call surface of the reference's
08_grid_and_montecarlo_localization/robot.py (class Robot, Table 8.1 of
Probabilistic Robotics).

What runs where.  The belief update -- `predict` (motion "convolution" over the
pose grid) and `update` (sector range query + beam likelihood for every cell and
heading, float32 normalisation) -- runs on the GPU through libmclhip.so
(`mclhip_grid_*`).  What concerns the ONE true pose stays on the host, written
with the same numpy scalar types as the reference so that a seeded run consumes
np.random identically and sees the same measured ranges: advancing the true
pose (robot.py:133-144), its true ranges (robot.py:226-275; a float32 pose makes
this float32 geometry) and drawing the noisy measurement from the tabulated
beam model (range_sensor.py:69-126).

There is no CPU fallback for the belief update: without the library or a GPU
the constructor raises.
"""
import numpy as np

from . import _lib
from .range_sensor import RangeSensor

res_x = 0.15        # robot.py:6-8
res_y = 0.15
res_theta = 90      # degrees per heading bin

max_x = 15
max_y = 15
w_map = int(max_x / res_x)
h_map = int(max_y / res_y)

max_sensor_range = 5
dt = 0.1
alpha = [0.2, 0.1, 0.2, 0.2, 0.1, 0.1]
pixel_obstacle = 200


class Robot:
    """Same constructor, attributes and methods as robot.py:22-342.

    Extra keyword arguments: `res_theta` (degrees per heading bin; the module
    constant by default), `sensor_angles`, `device`.  With more than four
    headings every heading is a neighbour in `predict` (the reference hard-codes
    its four, robot.py:168).
    """

    def __init__(self, map, x_pose, res_theta=None, sensor_angles=None, device=0):
        self.x = x_pose
        self.x_gt = x_pose.copy()
        self.range_sensor = RangeSensor(device)
        self.range_sensor.z_max = max_sensor_range
        self.sensor_angles = list(sensor_angles) if sensor_angles is not None else [
            0, 15, 30, 45, 60, 75, 90, 360 - 15, 360 - 30, 360 - 45, 360 - 60, 360 - 75, 360 - 90]
        n = len(self.sensor_angles)
        self.sensor_ranges = [0] * n
        self.true_ranges = [0] * n
        self.obstacles = []
        self.map = map
        self.res_theta = globals()["res_theta"] if res_theta is None else res_theta
        self.h_map, self.w_map = self.map.shape
        self.dir_map = int(360 // self.res_theta)

        self._ctx = _lib.Context(device, _lib.F64)
        self._ctx.grid_set_map(self.map, res_x, res_y, pixel_obstacle)
        rs = self.range_sensor
        self._ctx.grid_set_sensor(self.sensor_angles, self.dir_map, self.res_theta, rs.z_max, rs.sigma_hit,
                                  rs.lambda_short, [rs.w_hit, rs.w_short, rs.w_max, rs.w_rand], 5 * np.pi / 180.0)
        self._ctx.grid_init_belief()
        self._pk = None           # host mirrors, fetched on demand
        self._pk_pred = None
        self._pk_host_is_new = False
        self._pred_host_is_new = False
        # robot.py:49-60: before the first predict() pk_pred is the uniform prior too (float32)
        self._pk_pred = self.pk.copy()

    # -- belief mirrors -----------------------------------------------------
    @property
    def pk(self):
        """float32 [h_map, w_map, dir_map], as the reference holds it (robot.py:60, :300)."""
        if self._pk is None:
            self._pk = self._ctx.grid_get_belief(True, False)[0]
        return self._pk

    @pk.setter
    def pk(self, value):
        self._pk = np.ascontiguousarray(value, np.float32)
        self._ctx.grid_set_belief(self._pk)

    @property
    def pk_pred(self):
        if self._pk_pred is None:
            self._pk_pred = self._ctx.grid_get_belief(False, True)[1]
        return self._pk_pred

    @pk_pred.setter
    def pk_pred(self, value):
        self._pk_pred = np.ascontiguousarray(value, np.float64)
        self._ctx.grid_set_prediction(self._pk_pred)

    # -- Table 5.2 / 5.1 on the host (API surface; the GPU evaluates its own table) ----
    def prob_normal_distribution(self, a, b):
        return 1 / np.sqrt(2 * np.pi * b) * np.exp(-1 / 2 * (a ** 2) / b)

    def prob_triangular_distribution(self, a, b):
        prob = (np.sqrt(6 * b) - abs(a)) / (6 * b)
        prob *= abs(a) <= np.sqrt(6 * b)
        return prob

    def motion_model_velocity(self, xt, ut, xt_1, dt=1 / 10):
        """p(xt | ut, xt_1) for rows of xt_1 (robot.py:76-122)."""
        x1, y1, ha1 = xt
        xt_1 = np.asarray(xt_1)
        x0, y0, ha0 = xt_1[:, 0], xt_1[:, 1], xt_1[:, 2]
        c, s = np.cos(ha0), np.sin(ha0)
        t1 = (x0 - x1) * c + (y0 - y1) * s
        t2 = (y0 - y1) * c - (x0 - x1) * s
        t2[np.where(abs(t2) < 1e-6)] = 1e-6
        mu = 1.0 / 2.0 * t1 / t2
        cx = (x0 + x1) / 2 + mu * (y0 - y1)
        cy = (y0 + y1) / 2 + mu * (x1 - x0)
        cr = np.sqrt((x0 - cx) ** 2 + (y0 - cy) ** 2)
        delta_ha = np.arctan2(y1 - cy, x1 - cx) - np.arctan2(y0 - cy, x0 - cx)
        v_hat = delta_ha / dt * cr
        w_hat = delta_ha / dt
        gamma_hat = (ha1 - ha0) / dt - w_hat
        v, w = ut
        p = self.prob_normal_distribution
        return (p(v - v_hat, alpha[0] * abs(v) + alpha[1] * abs(w)) *
                p(w - w_hat, alpha[2] * abs(v) + alpha[3] * abs(w)) *
                p(gamma_hat, alpha[4] * abs(v) + alpha[5] * abs(w)))

    # -- filter steps ---------------------------------------------------------
    def predict(self, ut):
        """robot.py:125-190: the true pose moves on the host; the belief is predicted on the GPU."""
        vt, wt = ut
        x_gt, y_gt, ha_gt = self.x_gt[0, 0], self.x_gt[1, 0], self.x_gt[2, 0]
        if abs(wt) > 0:
            self.x_gt[0, 0] = x_gt - vt / wt * np.sin(ha_gt) + vt / wt * np.sin(ha_gt + wt * dt)
            self.x_gt[1, 0] = y_gt + vt / wt * np.cos(ha_gt) - vt / wt * np.cos(ha_gt + wt * dt)
            self.x_gt[2, 0] = ha_gt + wt * dt
        else:
            self.x_gt[0, 0] = x_gt + vt * np.cos(ha_gt) * dt
            self.x_gt[1, 0] = y_gt + vt * np.sin(ha_gt) * dt
            self.x_gt[2, 0] = ha_gt
        self._ctx.grid_predict(vt, wt, dt, alpha)
        self._pk_pred = None

    def update(self):
        """robot.py:277-310: measure at the true pose (host, consumes np.random), then the
        measurement update and float32 normalisation of the whole belief on the GPU."""
        z_meas = self.measure(self.x_gt, self.map)
        self._ctx.grid_update([float(z) for z in z_meas])
        self._pk = None

    # -- the sensor at ONE pose (host) -------------------------------------------
    def reset_obstacle_lut(self):
        self.obstacles = []

    def make_obstacle_lut(self, map):
        """robot.py:195-205: cells below the obstacle threshold, float32 metres."""
        if len(self.obstacles) == 0:
            rows, cols = np.nonzero(np.asarray(map) < pixel_obstacle)
            obs = np.stack([cols, rows], 1).astype(np.float32)
            if len(obs) > 0:
                obs[:, 0] *= res_x
                obs[:, 1] *= res_y
            self.obstacles = obs

    def measure(self, xt, map, noise=True):
        z_true = self.predict_measure(xt, map)
        for k in range(len(self.sensor_angles)):
            if noise:
                self.sensor_ranges[k] = self._draw(self._tabulate(z_true[k]))
            else:
                self.sensor_ranges[k] = z_true[k]
        return self.sensor_ranges

    def predict_measure(self, xt, map):
        """Nearest obstacle centre per +-5 degree sector (robot.py:226-275).  The dtype of xt
        carries through the arithmetic as in the reference."""
        x, y, theta = xt[0, 0], xt[1, 0], xt[2, 0]
        self.make_obstacle_lut(map)
        n = len(self.sensor_angles)
        if len(self.obstacles) == 0:
            for k in range(n):
                self.true_ranges[k] = max_sensor_range
            return self.true_ranges
        dx = self.obstacles[:, 0] - x
        dy = self.obstacles[:, 1] - y
        near = np.where(dx ** 2 + dy ** 2 <= max_sensor_range ** 2)
        dx, dy = dx[near], dy[near]
        bearing = np.arctan2(dy, dx)
        bearing[np.where(bearing < 0.0)] += np.pi * 2.0
        half = 5 * np.pi / 180.0
        for k in range(n):
            beam = self.sensor_angles[k] * np.pi / 180.0 + theta
            off = abs(beam - bearing)
            inside = np.where((off <= half) | ((off >= np.pi * 2 - half) & (off <= np.pi * 2 + half)))
            r2 = dx[inside] ** 2 + dy[inside] ** 2
            self.true_ranges[k] = np.sqrt(r2.min() if len(r2) > 0 else max_sensor_range ** 2)
        return self.true_ranges

    def _mixture(self, x, z):
        """range_sensor.py:95-112 on host scalars, keeping their numpy types (a float32 true range
        makes parts of this float32, as in the reference)."""
        rs = self.range_sensor
        p = 0
        if 0 <= x <= rs.z_max:
            s = rs.sigma_hit
            p = 1 / np.sqrt(2 * np.pi * s ** 2) * np.exp(-1 / 2 * (x - z) ** 2 / s ** 2) * rs.w_hit
        else:
            p = 0 * rs.w_hit
        if 0 <= x <= z:
            lam = rs.lambda_short
            eta = 1 / (1 - np.exp(-lam * z))
            p = p + eta * lam * np.exp(-lam * x) * rs.w_short
        else:
            p = p + 0 * rs.w_short
        p = p + (1 if abs(x - rs.z_max) < 1e-6 else 0) * rs.w_max
        p = p + (1 / rs.z_max if 0 <= x <= rs.z_max else 0) * rs.w_rand
        return p

    def _tabulate(self, z):
        rs = self.range_sensor
        n = int(rs.z_max / rs.resolution) + 1
        pdf = np.zeros(n)
        for i in range(n):
            pdf[i] = self._mixture(i * rs.resolution, z)
        rs.pdf = pdf
        return pdf

    def _draw(self, pdf):
        rs = self.range_sensor
        r = np.random.random() * pdf.sum()
        acc = 0
        for i in range(len(pdf)):
            if acc >= r:
                return i * rs.resolution
            acc += pdf[i]
        return 0

    def plot(self, sensor_readings=True):
        import matplotlib.pyplot as plt
        x, y, theta = self.x_gt[0, 0], self.x_gt[1, 0], self.x_gt[2, 0]
        plt.scatter(x / res_x, y / res_y, c='g')
        plt.plot([x / res_x, (x + np.cos(theta)) / res_x], [y / res_y, (y + np.sin(theta)) / res_y], c='g')
        if sensor_readings:
            for k, a in enumerate(self.sensor_angles):
                b = a * np.pi / 180.0 + theta
                r = self.sensor_ranges[k]
                plt.plot([x / res_x, (x + r * np.cos(b)) / res_x], [y / res_y, (y + r * np.sin(b)) / res_y], c='r')
