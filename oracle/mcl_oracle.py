"""CPU oracle for the Monte Carlo localization update  --  TEST INFRASTRUCTURE ONLY.

This module is a numpy (float64) restatement of the reference algorithm, stage
by stage.  It exists so that tests, the smoke check and the `cpu_baseline` leg
of the benchmark have something to compare the HIP path with.  Nothing in the
product package imports it; the product path never falls back to it.

Parity status: PINNED.  Every stage below reproduces, bit for bit on the
platform that generated them, the golden vectors under tests/golden/ that
tools/gen_golden.py captured by importing the reference itself
(tests/test_oracle_golden.py).

Reference files (paths relative to the reference checkout):
  08_grid_and_montecarlo_localization/mc_localization.py   (MCL, Table 8.2)
  08_grid_and_montecarlo_localization/range_sensor.py      (beam model, Eq. 6.13)
  04_nonparametric_filters/particle_filter.py:58-81        (low-variance sampler)
  05_robot_motion/sample_motion_model_velocity.py:73-108   (Tables 5.3 / 5.4)
"""
import math

import numpy as np

# module constants of mc_localization.py:9-18
RES_X = 0.15
RES_Y = 0.15
MAX_SENSOR_RANGE = 5
ALPHA = (4, 2, 4, 4, 4, 2)
SECTOR_HALF_ANGLE = 5 * np.pi / 180.0  # mc_localization.py:139
SUM_CHUNK = 8192  # numpy's reduction buffer: np.sum accumulates chunk by chunk


class SensorParams:
    """Constants of RangeSensor.__init__ (range_sensor.py:5-28) with z_max
    overridden to max_sensor_range as mc_localization.py:55-56 does."""

    def __init__(self, z_max=MAX_SENSOR_RANGE, sigma_hit=0.15, lambda_short=1.0,
                 w_hit=0.97, w_short=0.005, w_max=0.020, w_rand=0.005):
        s = w_hit + w_short + w_max + w_rand  # range_sensor.py:22-26
        self.w_hit = w_hit / s
        self.w_short = w_short / s
        self.w_max = w_max / s
        self.w_rand = w_rand / s
        self.sigma_hit = sigma_hit
        self.lambda_short = lambda_short
        self.z_max = z_max


# --------------------------------------------------------------------------
# a1  noise sampler                                  mc_localization.py:20-24
# --------------------------------------------------------------------------
def sum12(ints):
    """Sum of 12 scaled integers in the order numpy's pairwise reduction uses
    for n = 12: eight strided accumulators combined as a tree, then a serial
    tail.  ints: [..., 12] integer array.  Returns float64 [...]."""
    r = ints.astype(np.int64) * 0.001
    a = (r[..., 0] + r[..., 1]) + (r[..., 2] + r[..., 3])
    b = (r[..., 4] + r[..., 5]) + (r[..., 6] + r[..., 7])
    s = a + b
    for i in range(8, 12):
        s = s + r[..., i]
    return s


def noise_coeffs(ut, alpha=ALPHA):
    """b/6.0 for the three draws of one motion sample (mc_localization.py:89-91,
    :24).  Python-float arithmetic, exactly as the reference evaluates it."""
    vt, wt = ut
    b1 = alpha[0] * abs(vt) + alpha[1] * abs(wt)
    b2 = alpha[2] * abs(vt) + alpha[3] * abs(wt)
    b3 = alpha[4] * abs(vt) + alpha[5] * abs(wt)
    return b1 / 6.0, b2 / 6.0, b3 / 6.0


# --------------------------------------------------------------------------
# a2  velocity motion model                         mc_localization.py:74-114
# --------------------------------------------------------------------------
def free_cell(x, y, grid, res_x=RES_X, res_y=RES_Y):
    """pm of mc_localization.py:106-112: 1.0 where int(x/res+0.5) (truncation
    toward zero) lands on a cell whose value is exactly 255."""
    u = np.trunc(x / res_x + 0.5)
    v = np.trunc(y / res_y + 0.5)
    H, W = grid.shape
    inb = (u >= 0) & (u < W) & (v >= 0) & (v < H)
    ui = np.where(inb, u, 0).astype(np.int64)
    vi = np.where(inb, v, 0).astype(np.int64)
    return np.where(inb & (grid[vi, ui] == 255), 1.0, 0.0)


def motion(X, ut, dt, grid, noise_ints=None, alpha=ALPHA, res_x=RES_X, res_y=RES_Y):
    """Table 5.3 over a batch.  X [M,3] float64; noise_ints [M,3,12] or None
    (noise-free).  Returns (X_new [M,3], pm [M])."""
    X = np.asarray(X, np.float64)
    vt, wt = ut
    x, y, th = X[:, 0], X[:, 1], X[:, 2]
    if noise_ints is not None:
        c1, c2, c3 = noise_coeffs(ut, alpha)
        s = sum12(np.asarray(noise_ints))
        v_hat = vt + c1 * s[:, 0]
        w_hat = wt + c2 * s[:, 1]
        g_hat = c3 * s[:, 2]
    else:
        v_hat = np.full(len(X), float(vt))
        w_hat = np.full(len(X), float(wt))
        g_hat = np.zeros(len(X))
    out = np.zeros_like(X)
    arc = np.abs(w_hat) > 0
    with np.errstate(divide="ignore", invalid="ignore"):
        q = v_hat / w_hat
        ax = x - q * np.sin(th) + q * np.sin(th + w_hat * dt)
        ay = y + q * np.cos(th) - q * np.cos(th + w_hat * dt)
        at = th + w_hat * dt + g_hat * dt
    sx = x + v_hat * np.cos(th) * dt
    sy = y + v_hat * np.sin(th) * dt
    st = th + g_hat * dt
    out[:, 0] = np.where(arc, ax, sx)
    out[:, 1] = np.where(arc, ay, sy)
    out[:, 2] = np.where(arc, at, st)
    return out, free_cell(out[:, 0], out[:, 1], grid, res_x, res_y)


ALPHA_BOOK = (0.2, 0.1, 0.2, 0.2, 0.1, 0.1)   # 05_robot_motion/sample_motion_model_velocity.py:11


def motion_book(X, ut, dt, noise_ints=None, alpha=ALPHA_BOOK):
    """Table 5.3 as the chapter-5 script has it (05_robot_motion/sample_motion_model_velocity.py:90-108):
    the arc form for every w_hat -- w_hat == 0 divides by zero in float64 (inf / NaN poses) -- and no
    map.  X [M,3]; noise_ints [M,3,12] or None.  Returns X_new [M,3]."""
    X = np.asarray(X, np.float64)
    vt, wt = float(ut[0]), float(ut[1])
    x, y, th = X[:, 0], X[:, 1], X[:, 2]
    if noise_ints is not None:
        c1, c2, c3 = noise_coeffs((vt, wt), alpha)
        s = sum12(np.asarray(noise_ints))
        v_hat = vt + c1 * s[:, 0]          # :96-98
        w_hat = wt + c2 * s[:, 1]
        g_hat = c3 * s[:, 2]
    else:
        v_hat = np.full(len(X), vt)
        w_hat = np.full(len(X), wt)
        g_hat = np.zeros(len(X))
    out = np.zeros_like(X)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = v_hat / w_hat
        out[:, 0] = x - q * np.sin(th) + q * np.sin(th + w_hat * dt)   # :102
        out[:, 1] = y + q * np.cos(th) - q * np.cos(th + w_hat * dt)   # :103
        out[:, 2] = th + w_hat * dt + g_hat * dt                       # :104
    return out


def motion_single(xt_1, ut, dt, grid, res_x=RES_X, res_y=RES_Y):
    """Noise-free single-pose update as sample() applies it to the true pose
    (mc_localization.py:223).  Keeps the dtype of xt_1's elements in the
    arithmetic: the demo's first true pose is float32 (mc_localization.py:306)."""
    vt, wt = ut
    x, y, th = xt_1[0], xt_1[1], xt_1[2]
    out = np.zeros([3])
    if abs(wt) > 0:
        out[0] = x - vt / wt * np.sin(th) + vt / wt * np.sin(th + wt * dt)
        out[1] = y + vt / wt * np.cos(th) - vt / wt * np.cos(th + wt * dt)
        out[2] = th + wt * dt + 0 * dt
    else:
        out[0] = x + vt * np.cos(th) * dt
        out[1] = y + vt * np.sin(th) * dt
        out[2] = th + 0 * dt
    pm = free_cell(out[0:1], out[1:2], grid, res_x, res_y)[0]
    return out, pm


# --------------------------------------------------------------------------
# a3  obstacle look-up table                          mc_localization.py:61-71
# --------------------------------------------------------------------------
def obstacle_lut(grid, res_x=RES_X, res_y=RES_Y):
    """Cells equal to 0 in row-major order; coordinates are float32 products
    float32(col)*float32(res_x), float32(row)*float32(res_y)."""
    rows, cols = np.nonzero(grid == 0)
    lut = np.stack([cols, rows], 1).astype(np.float32)
    if len(lut):
        lut[:, 0] *= np.float32(res_x)
        lut[:, 1] *= np.float32(res_y)
    cell = rows.astype(np.int64) * grid.shape[1] + cols
    return lut, cell


# --------------------------------------------------------------------------
# a4  sector range query                            mc_localization.py:116-171
# --------------------------------------------------------------------------
def sector_query(pose, lut, cell, angles_deg, z_max=MAX_SENSOR_RANGE, half_angle=SECTOR_HALF_ANGLE):
    """Ranges [B] and hit cells [B] (row-major index of the nearest obstacle
    cell centre inside each beam's sector, lowest index on ties; -1 = none)."""
    B = len(angles_deg)
    r2max = z_max ** 2
    ranges = np.full(B, np.sqrt(np.float64(r2max)))
    hits = np.full(B, -1, np.int64)
    if len(lut) == 0:
        ranges[:] = z_max
        return ranges, hits
    x, y, th = np.float64(pose[0]), np.float64(pose[1]), np.float64(pose[2])
    dx = lut[:, 0] - x  # float32 array - float64 scalar -> float64 (NEP 50)
    dy = lut[:, 1] - y
    r2 = dx ** 2 + dy ** 2
    near = np.nonzero(r2 <= r2max)[0]
    dx, dy, r2 = dx[near], dy[near], r2[near]
    ang = np.arctan2(dy, dx)
    ang[ang < 0.0] += np.pi * 2.0
    two_pi_lo = np.pi * 2 - half_angle
    two_pi_hi = np.pi * 2 + half_angle
    for k in range(B):
        beam = angles_deg[k] * np.pi / 180.0 + th  # never wrapped
        d = np.abs(beam - ang)
        sel = np.nonzero((d <= half_angle) | ((d >= two_pi_lo) & (d <= two_pi_hi)))[0]
        if len(sel):
            j = sel[np.argmin(r2[sel])]
            ranges[k] = np.sqrt(r2[j])
            hits[k] = cell[near[j]]
    return ranges, hits


def sector_query_batch(X, lut, cell, angles_deg, z_max=MAX_SENSOR_RANGE, half_angle=SECTOR_HALF_ANGLE):
    M, B = len(X), len(angles_deg)
    R = np.zeros((M, B))
    Hc = np.zeros((M, B), np.int64)
    for m in range(M):
        R[m], Hc[m] = sector_query(X[m], lut, cell, angles_deg, z_max, half_angle)
    return R, Hc


# --------------------------------------------------------------------------
# a5  beam mixture                                    range_sensor.py:31-112
# --------------------------------------------------------------------------
def _scalar_square(d):
    """d**2 as the reference evaluates it: on numpy float64 SCALARS `**`
    goes through libm pow(), which is not always equal to d*d (about 1 value in
    2000 differs by one ulp), so the square is taken element by element."""
    flat = np.asarray(d, np.float64).ravel()
    out = np.fromiter((math.pow(v, 2.0) for v in flat), np.float64, len(flat))
    return out.reshape(np.shape(d))


def beam_px(z, z_true, sp):
    """p(z | z*) with z = the particle's predicted range and z_true = the
    measured range (argument roles of mc_localization.py:237).  Arrays broadcast."""
    z = np.asarray(z, np.float64)
    zt = np.asarray(z_true, np.float64)
    in_range = (0 <= z) & (z <= sp.z_max)
    s = sp.sigma_hit
    p_hit = np.where(in_range, 1 / np.sqrt(2 * np.pi * s ** 2) * np.exp(-1 / 2 * _scalar_square(z - zt) / s ** 2), 0.0)
    lam = sp.lambda_short
    with np.errstate(divide="ignore", invalid="ignore"):
        eta = 1 / (1 - np.exp(-lam * zt))
        p_short = np.where((0 <= z) & (z <= zt), eta * lam * np.exp(-lam * z), 0.0)
    p_max = np.where(np.abs(z - sp.z_max) < 1e-6, 1.0, 0.0)
    p_rand = np.where(in_range, 1 / sp.z_max, 0.0)
    return p_hit * sp.w_hit + p_short * sp.w_short + p_max * sp.w_max + p_rand * sp.w_rand


def weights_from_ranges(R, z_meas, pm, sp):
    """q = prod_k p_k accumulated left to right from 1.0, times pm
    (mc_localization.py:235-240).  Also returns log-weights sum_k log p_k
    (-inf where pm == 0)."""
    P = beam_px(R, np.asarray(z_meas)[None, :], sp)
    q = np.ones(len(R))
    for k in range(P.shape[1]):
        q = q * P[:, k]
    with np.errstate(divide="ignore"):
        logw = np.sum(np.log(P), axis=1) + np.where(pm > 0, 0.0, -np.inf)
    return q * pm, logw, P


# --------------------------------------------------------------------------
# a6  normalisation                                   mc_localization.py:248
# --------------------------------------------------------------------------
def _pairwise(a):
    n = len(a)
    if n < 8:
        s = 0.0
        for v in a:
            s = s + v
        return s
    if n <= 128:
        acc = [a[i] for i in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                acc[j] = acc[j] + a[i + j]
            i += 8
        s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]))
        while i < n:
            s = s + a[i]
            i += 1
        return s
    h = n // 2
    h -= h % 8
    return _pairwise(a[:h]) + _pairwise(a[h:])


def numpy_order_sum(w):
    """np.sum(w) for a contiguous float64 vector restated explicitly: chunks of
    8192 elements are reduced by numpy's pairwise scheme and the chunk results
    are accumulated left to right."""
    w = np.asarray(w, np.float64)
    total = None
    for s in range(0, len(w), SUM_CHUNK):
        c = _pairwise([np.float64(v) for v in w[s:s + SUM_CHUNK]])
        total = c if total is None else total + c
    return np.float64(0.0) if total is None else np.float64(total)


def normalise(w):
    with np.errstate(divide="ignore", invalid="ignore"):
        return w / np.sum(w)


# --------------------------------------------------------------------------
# a7  low-variance resampling                       mc_localization.py:242-262
# --------------------------------------------------------------------------
def resample_indices_loop(w_norm, r):
    """The reference's two-pointer walk, verbatim in structure (small M only)."""
    M = len(w_norm)
    c = w_norm[0]
    i = 0
    out = np.zeros(M, np.int64)
    for m in range(M):
        U = r + (m / M)
        while U > c:
            i = i + 1
            c = c + w_norm[i]  # IndexError past the end, as in the reference
        out[m] = i
    return out


def resample_indices(w_norm, r, clamp=False):
    """Same indices without the Python loop: the running sum c is the strictly
    sequential float64 chain (== np.cumsum), U_m = r + m/M, index = number of
    c_i that are < U_m.  With clamp=False an index past the end raises
    IndexError like the reference's unguarded loop; clamp=True returns M-1
    (the guard 13_fast_slam/fast_slam1.py:426 applies)."""
    w_norm = np.asarray(w_norm, np.float64)
    M = len(w_norm)
    c = np.cumsum(w_norm)
    U = r + (np.arange(M) / M)
    idx = np.searchsorted(c, U, side="left")
    if np.isnan(c[-1]):
        # NaN weights: `U > c` is always False, the walk never advances
        return np.zeros(M, np.int64)
    if idx.max() >= M:
        if not clamp:
            raise IndexError("resample walk ran past the last particle")
        idx = np.minimum(idx, M - 1)
    return idx.astype(np.int64)


# --------------------------------------------------------------------------
# a8  one MCL update                                mc_localization.py:221-262
# --------------------------------------------------------------------------
class OracleMCL:
    def __init__(self, grid, sensor_angles, x_gt, sp=None, res_x=RES_X, res_y=RES_Y, alpha=ALPHA):
        self.grid = np.asarray(grid)
        self.angles = list(sensor_angles)
        self.sp = sp or SensorParams()
        self.res_x, self.res_y, self.alpha = res_x, res_y, alpha
        self.lut, self.cell = obstacle_lut(self.grid, res_x, res_y)
        self.x_gt = x_gt
        self.X = None
        self.w = None
        self.z_meas = None

    def sample(self, ut, dt, noise_ints):
        self.x_gt, _ = motion_single(self.x_gt, ut, dt, self.grid, self.res_x, self.res_y)
        self.z_meas, self.hits_gt = sector_query(self.x_gt, self.lut, self.cell, self.angles, self.sp.z_max)
        self.X, self.pm = motion(self.X, ut, dt, self.grid, noise_ints, self.alpha, self.res_x, self.res_y)
        self.R, self.hits = sector_query_batch(self.X, self.lut, self.cell, self.angles, self.sp.z_max)
        self.w, self.logw, self.P = weights_from_ranges(self.R, self.z_meas, self.pm, self.sp)

    def resample(self, r, clamp=False):
        self.w = normalise(self.w)
        self.idx = resample_indices(self.w, r, clamp)
        self.X = self.X[self.idx]
        self.w = self.w[self.idx]


# --------------------------------------------------------------------------
# next row: augmented MCL                 mc_localization2.py (Table 8.3)
# --------------------------------------------------------------------------
def amcl_weights(R, range_noise, z_meas, pm, sp, no_obstacles=False):
    """Weights with noisy predicted ranges: r += randn()*0.15 per beam
    (mc_localization2.py:166-168); without obstacles the reading is
    min(z_max, z_max + noise) (:170-173).  Returns (w, P)."""
    Z = np.minimum(sp.z_max, sp.z_max + range_noise) if no_obstacles else R + range_noise
    P = beam_px(Z, np.asarray(z_meas)[None, :], sp)
    q = np.ones(len(Z))
    for k in range(P.shape[1]):
        q = q * P[:, k]
    return q * pm, P


def sequential_mean(w):
    """w_avg of mc_localization2.py:228-246: `w_avg += w[m] / M` from 0, strictly left to right."""
    w = np.asarray(w, np.float64)
    return np.float64(np.cumsum(w / len(w))[-1])


def amcl_rates(w_slow, w_fast, w_avg, a_slow=0.001, a_fast=0.9):
    """mc_localization2.py:248-249."""
    return w_slow + a_slow * (w_avg - w_slow), w_fast + a_fast * (w_avg - w_fast)


def amcl_resample(X, w_norm, r, inject_rows, inject_poses, clamp=False):
    """mc_localization2.py:266-296.  Injected positions get a fresh pose and weight 1/M; the
    others follow the low-variance walk.  The reference writes the injected poses into self.X
    in place and collects row *views*, so every position that resamples an injected row ends
    up with the injected pose as well -- reproduced here."""
    M = len(w_norm)
    idx = resample_indices(w_norm, r, clamp=True)
    inject_rows = np.asarray(inject_rows, np.int64)
    walk = np.ones(M, bool)
    walk[inject_rows] = False
    c = np.cumsum(w_norm)
    if not clamp and np.any(walk) and not np.isnan(c[-1]):
        U = r + (np.arange(M) / M)
        if np.any(U[walk] > c[-1]):
            raise IndexError("resample walk ran past the last particle")
    Xmod = np.array(X, np.float64, copy=True)
    if len(inject_rows):
        Xmod[inject_rows] = inject_poses
    Xo = Xmod[idx]
    wo = np.asarray(w_norm, np.float64)[idx].copy()
    if len(inject_rows):
        Xo[inject_rows] = inject_poses
        wo[inject_rows] = 1.0 / M
    return Xo, wo, idx
