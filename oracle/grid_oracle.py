"""CPU oracle for grid (histogram) localisation  --  TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's Table 8.1 filter, stage by stage.  Tests and
the smoke check compare the HIP path with it; nothing in the product package
imports it and the product path never falls back to it.

Parity status: PINNED for the reference's own configuration (100x100 map, four
headings, 13 beams): tools/gen_golden.py imports the reference's Robot class,
runs predict()/update() under a seeded np.random and stores every stage's input
and output (tests/golden/grid_*.npz); tests/test_oracle_golden.py checks this
module against them.  For more than four headings the reference has no
behaviour of its own (its neighbour list is hard-coded to headings 0..3,
robot.py:168); this module then sums over all headings -- parity unpinned for
that generalisation.

Reference files (paths relative to the reference checkout):
  08_grid_and_montecarlo_localization/robot.py            (Robot: predict, update, measure)
  08_grid_and_montecarlo_localization/range_sensor.py     (beam model, pdf sampling)
  08_grid_and_montecarlo_localization/grid_localization.py (driver)
"""
import math

import numpy as np

from . import mcl_oracle as M

RES_X = 0.15                 # robot.py:6-7
RES_Y = 0.15
RES_THETA = 90               # robot.py:8 (degrees per heading bin)
MAX_SENSOR_RANGE = 5         # robot.py:16
DT = 0.1                     # robot.py:17
ALPHA = (0.2, 0.1, 0.2, 0.2, 0.1, 0.1)   # robot.py:18
PIXEL_OBSTACLE = 200         # robot.py:20
SENSOR_ANGLES = (0, 15, 30, 45, 60, 75, 90, 360 - 15, 360 - 30, 360 - 45, 360 - 60, 360 - 75, 360 - 90)  # robot.py:31-33
REACH = 3                    # neighbourhood half-width of predict(), robot.py:161


def initial_belief(grid, n_dir):
    """robot.py:49-60: uniform over cells whose map value is not 0, float32."""
    pk = np.ones([grid.shape[0], grid.shape[1], n_dir], dtype=np.float32)
    pk[grid == 0, :] = 0
    pk /= int(np.count_nonzero(grid != 0))
    return pk


def heading(d, res_theta=RES_THETA):
    """d*res_theta*np.pi/180.0 with Python's left-to-right evaluation (robot.py:159, :178)."""
    return d * res_theta * np.pi / 180.0


def prob_normal(a, b):
    """robot.py:62-65 on an array a and a Python-float variance b."""
    return 1 / np.sqrt(2 * np.pi * b) * np.exp(-1 / 2 * (a ** 2) / b)


def motion_density(x1, y1, ha1, ut, x0, y0, ha0, dt=DT, alpha=ALPHA):
    """p(xt | ut, xt-1) of robot.py:76-122 for arrays of previous poses (x0, y0, ha0) and a
    target pose (x1, y1, ha1) that broadcasts against them."""
    cos_ha0 = np.cos(ha0)
    sin_ha0 = np.sin(ha0)
    t1 = (x0 - x1) * cos_ha0 + (y0 - y1) * sin_ha0
    t2 = (y0 - y1) * cos_ha0 - (x0 - x1) * sin_ha0
    t2 = np.where(np.abs(t2) < 1e-6, 1e-6, t2)            # :97-98
    mu = 1.0 / 2.0 * t1 / t2                               # :100
    mx = (x0 + x1) / 2
    my = (y0 + y1) / 2
    cx = mx + mu * (y0 - y1)                               # :106
    cy = my + mu * (x1 - x0)                               # :107
    cr = np.sqrt((x0 - cx) ** 2 + (y0 - cy) ** 2)          # :108
    delta_ha = np.arctan2(y1 - cy, x1 - cx) - np.arctan2(y0 - cy, x0 - cx)   # :109
    v_hat = delta_ha / dt * cr                             # :110
    w_hat = delta_ha / dt                                  # :111
    gamma_hat = (ha1 - ha0) / dt - w_hat                   # :112
    v, w = ut
    pv = prob_normal(v - v_hat, alpha[0] * abs(v) + alpha[1] * abs(w))
    pw = prob_normal(w - w_hat, alpha[2] * abs(v) + alpha[3] * abs(w))
    pg = prob_normal(gamma_hat, alpha[4] * abs(v) + alpha[5] * abs(w))
    return pv * pw * pg


def pairwise_rows(A):
    """numpy's pairwise summation (the order np.sum uses on a contiguous 1-D vector shorter than
    its 8192-element buffer) applied to every row of A at once."""
    n = A.shape[1]
    if n < 8:
        s = np.zeros(A.shape[0], A.dtype)
        for i in range(n):
            s = s + A[:, i]
        return s
    if n <= 128:
        acc = [A[:, j].copy() for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                acc[j] = acc[j] + A[:, i + j]
            i += 8
        s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]))
        while i < n:
            s = s + A[:, i]
            i += 1
        return s
    h = n // 2
    h -= h % 8
    return pairwise_rows(A[:, :h]) + pairwise_rows(A[:, h:])


def predict(pk, grid, ut, n_dir, dt=DT, alpha=ALPHA, res_x=RES_X, res_y=RES_Y, res_theta=RES_THETA,
            neighbour_dirs=None):
    """robot.py:148-187.  pk float32 [H,W,D] -> pk_pred float64 [H,W,D]: for interior cells
    (three cells off every border) whose map value is >= 200, the sum over the 7x7xD
    neighbourhood of pk * p(xt | ut, xt-1); zero elsewhere.  The neighbour order is (row, col,
    heading) and the sum runs in numpy's pairwise order (:183-187)."""
    H, W = grid.shape
    nds = list(range(4)) if neighbour_dirs is None else list(neighbour_dirs)   # :168
    out = np.zeros([H, W, n_dir])
    ii, jj = np.nonzero(grid[REACH:H - REACH, REACH:W - REACH] >= PIXEL_OBSTACLE)
    ii = ii + REACH
    jj = jj + REACH
    if len(ii) == 0:
        return out
    oi, oj, od = np.meshgrid(np.arange(-REACH, REACH + 1), np.arange(-REACH, REACH + 1), np.array(nds), indexing="ij")
    oi, oj, od = oi.ravel(), oj.ravel(), od.ravel()        # neighbour order of :175-181
    ni = ii[:, None] + oi[None, :]
    nj = jj[:, None] + oj[None, :]
    x0 = nj * res_x
    y0 = ni * res_y
    ha0 = np.array([heading(int(d), res_theta) for d in od])[None, :]
    pt_1 = pk[ni, nj, od[None, :]]                         # float32
    x1 = (jj * res_x)[:, None]
    y1 = (ii * res_y)[:, None]
    for d in range(n_dir):
        p = motion_density(x1, y1, heading(d, res_theta), ut, x0, y0, ha0, dt, alpha)
        out[ii, jj, d] = pairwise_rows(pt_1 * p)
    return out


def advance_true_pose(x_gt, ut, dt=DT):
    """robot.py:133-144 on the [3,1] true pose, in place, with the array's own dtype deciding the
    arithmetic (the driver's pose is float32, grid_localization.py:25)."""
    vt, wt = ut
    x, y, ha = x_gt[0, 0], x_gt[1, 0], x_gt[2, 0]
    if abs(wt) > 0:
        x_gt[0, 0] = x - vt / wt * np.sin(ha) + vt / wt * np.sin(ha + wt * dt)
        x_gt[1, 0] = y + vt / wt * np.cos(ha) - vt / wt * np.cos(ha + wt * dt)
        x_gt[2, 0] = ha + wt * dt
    else:
        x_gt[0, 0] = x + vt * np.cos(ha) * dt
        x_gt[1, 0] = y + vt * np.sin(ha) * dt
        x_gt[2, 0] = ha
    return x_gt


def obstacle_lut(grid, res_x=RES_X, res_y=RES_Y):
    """robot.py:195-205: cells below 200, row-major, float32 coordinates; plus their flat index."""
    rows, cols = np.nonzero(grid < PIXEL_OBSTACLE)
    lut = np.stack([cols, rows], 1).astype(np.float32)
    if len(lut):
        lut[:, 0] *= res_x
        lut[:, 1] *= res_y
    return lut, rows.astype(np.int64) * grid.shape[1] + cols


def ranges_at(xt, lut, angles=SENSOR_ANGLES, z_max=MAX_SENSOR_RANGE):
    """robot.py:226-275 for one [3,1] pose; the pose's dtype decides the arithmetic exactly as in
    the reference (float32 pose -> float32 geometry, float64 pose -> float64)."""
    x, y, theta = xt[0, 0], xt[1, 0], xt[2, 0]
    out = [0] * len(angles)
    if len(lut) == 0:
        return [z_max] * len(angles)
    dx = lut[:, 0] - x
    dy = lut[:, 1] - y
    r2 = dx ** 2 + dy ** 2
    keep = np.where(r2 <= z_max ** 2)
    dx = dx[keep]
    dy = dy[keep]
    ang = np.arctan2(dy, dx)
    ang[np.where(ang < 0.0)] += np.pi * 2.0
    thres = 5 * np.pi / 180.0
    for k, a in enumerate(angles):
        beam = a * np.pi / 180.0 + theta
        diff = abs(beam - ang)
        sel = np.where((diff <= thres) | ((diff >= np.pi * 2 - thres) & (diff <= np.pi * 2 + thres)))
        rr = dx[sel] ** 2 + dy[sel] ** 2
        out[k] = np.sqrt(rr.min() if len(rr) > 0 else z_max ** 2)
    return out


class Sensor:
    """RangeSensor of range_sensor.py:4-29 with z_max = 5 (robot.py:28-29)."""

    def __init__(self):
        sp = M.SensorParams()
        self.w_hit, self.w_short, self.w_max, self.w_rand = sp.w_hit, sp.w_short, sp.w_max, sp.w_rand
        self.sigma_hit, self.lambda_short, self.z_max, self.resolution = 0.15, 1.0, 5, 0.01

    def p_hit(self, x, z):     # range_sensor.py:31-42
        if 0 <= x <= self.z_max:
            s = self.sigma_hit
            return 1 / np.sqrt(2 * np.pi * s ** 2) * np.exp(-1 / 2 * (x - z) ** 2 / s ** 2)
        return 0

    def p_short(self, x, z):   # range_sensor.py:44-53
        if 0 <= x <= z:
            lam = self.lambda_short
            eta = 1 / (1 - np.exp(-lam * z))
            return eta * lam * np.exp(-lam * x)
        return 0

    def p_max(self, x):        # range_sensor.py:55-60
        return 1 if abs(x - self.z_max) < 1e-6 else 0

    def p_rand(self, x):       # range_sensor.py:62-67
        return 1 / self.z_max if 0 <= x <= self.z_max else 0

    def px(self, z, z_true):   # range_sensor.py:95-112
        return (self.p_hit(z, z_true) * self.w_hit + self.p_short(z, z_true) * self.w_short +
                self.p_max(z) * self.w_max + self.p_rand(z) * self.w_rand)

    def pdf(self, z):          # range_sensor.py:69-93
        n = int(self.z_max / self.resolution) + 1
        out = np.zeros(n)
        for xi in range(n):
            out[xi] = self.px(xi * self.resolution, z)
        return out

    def draw(self, pdf, u):    # range_sensor.py:114-126 with u = np.random.random()
        r = u * pdf.sum()
        acc = 0
        for k in range(len(pdf)):
            if acc >= r:
                return k * self.resolution
            acc += pdf[k]
        return 0


def sense(x_gt, lut, uniforms, angles=SENSOR_ANGLES):
    """robot.py:208-224 with noise: true ranges at the (float32) true pose, then one draw per beam
    from the beam model's tabulated pdf.  uniforms: one np.random.random() value per beam."""
    s = Sensor()
    z_true = ranges_at(x_gt, lut, angles)
    return [s.draw(s.pdf(z_true[k]), uniforms[k]) for k in range(len(angles))], z_true


def cell_likelihood(z_meas, z_pred, sp=None):
    """q = prod_k compute_px(z_meas[k], z_pred[., k]) from 1.0, left to right (robot.py:294-298).
    The measured value is the first argument here (the other way round from the particle
    filter).  z_pred [N,B] float64, z_meas B Python floats.  Returns (q [N], P [N,B])."""
    sp = sp or M.SensorParams()
    z_pred = np.asarray(z_pred, np.float64)
    N, B = z_pred.shape
    P = np.zeros((N, B))
    s = sp.sigma_hit
    lam = sp.lambda_short
    for k in range(B):
        x = float(z_meas[k])
        z = z_pred[:, k]
        in_range = 0 <= x <= sp.z_max
        p_hit = (1 / np.sqrt(2 * np.pi * s ** 2) * np.exp(-1 / 2 * M._scalar_square(x - z) / s ** 2)) if in_range else 0.0
        with np.errstate(divide="ignore", invalid="ignore"):
            eta = 1 / (1 - np.exp(-lam * z))
            p_short = np.where(x <= z, eta * lam * np.exp(-lam * x), 0.0) if 0 <= x else 0.0
        p_max = 1 if abs(x - sp.z_max) < 1e-6 else 0
        p_rand = 1 / sp.z_max if in_range else 0
        P[:, k] = p_hit * sp.w_hit + p_short * sp.w_short + p_max * sp.w_max + p_rand * sp.w_rand
    q = np.ones(N)
    for k in range(B):
        q = q * P[:, k]
    return q, P


def f32_sum(a):
    """np.sum of a contiguous float32 array restated: 8192-element chunks, pairwise inside with
    float32 accumulators, chunk results added left to right in float32."""
    flat = np.ascontiguousarray(a, np.float32).ravel()
    total = None
    for s in range(0, len(flat), M.SUM_CHUNK):
        c = pairwise_rows(flat[None, s:s + M.SUM_CHUNK])[0]
        total = c if total is None else np.float32(total + c)
    return np.float32(0.0) if total is None else np.float32(total)


def predicted_ranges(grid, n_dir, angles=SENSOR_ANGLES, res_x=RES_X, res_y=RES_Y, res_theta=RES_THETA, z_max=MAX_SENSOR_RANGE):
    """z_pred for every cell with map >= 200 and every heading (robot.py:289-292): float64 poses
    (j*res_x, i*res_y, heading(d)).  Same arithmetic as ranges_at, with the part that does not
    depend on the heading (offsets, bearings) evaluated once per cell.
    Returns (rows, cols, Z [N, D, B])."""
    lut, cell = obstacle_lut(grid, res_x, res_y)
    ii, jj = np.nonzero(grid >= 200)
    B = len(angles)
    Z = np.full((len(ii), n_dir, B), np.sqrt(np.float64(z_max ** 2)))
    if len(lut) == 0:
        return ii, jj, Z
    thres = 5 * np.pi / 180.0
    for n in range(len(ii)):
        dx = lut[:, 0] - np.float64(jj[n] * res_x)
        dy = lut[:, 1] - np.float64(ii[n] * res_y)
        r2 = dx ** 2 + dy ** 2
        keep = np.where(r2 <= z_max ** 2)
        dx, dy, r2 = dx[keep], dy[keep], r2[keep]
        ang = np.arctan2(dy, dx)
        ang[np.where(ang < 0.0)] += np.pi * 2.0
        for d in range(n_dir):
            theta = np.float64(heading(d, res_theta))
            for k, a in enumerate(angles):
                diff = abs((a * np.pi / 180.0 + theta) - ang)
                sel = (diff <= thres) | ((diff >= np.pi * 2 - thres) & (diff <= np.pi * 2 + thres))
                if sel.any():
                    Z[n, d, k] = np.sqrt(r2[sel].min())
    return ii, jj, Z


def update(pk_pred, grid, z_meas, n_dir, angles=SENSOR_ANGLES, Z=None):
    """robot.py:277-310: pk = float32(pk_pred * q) on cells with map >= 200, 0 elsewhere, then
    pk /= pk.sum() in float32.  Returns (pk float32 [H,W,D], the float32 sum, unnormalised pk)."""
    H, W = grid.shape
    if Z is None:
        ii, jj, Z = predicted_ranges(grid, n_dir, angles)
    else:
        ii, jj = np.nonzero(grid >= 200)
    pk = np.zeros([H, W, n_dir], np.float32)
    for d in range(n_dir):
        q, _ = cell_likelihood(z_meas, Z[:, d, :])
        pk[ii, jj, d] = pk_pred[ii, jj, d] * q
    raw = pk.copy()
    total = f32_sum(pk)
    with np.errstate(divide="ignore", invalid="ignore"):
        pk /= total
    return pk, total, raw
