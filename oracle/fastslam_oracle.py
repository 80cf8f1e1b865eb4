"""CPU oracle for FastSLAM 1.0 with unknown correspondence  --  TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's filter step (Table 13.1 / 13.2 as the reference modifies them),
fed with pre-drawn random numbers instead of np.random so that the HIP path can consume the very
same stream.  Nothing in the product package imports it.

Parity status: PINNED.  tools/gen_golden.py (`fastslam`) drives the reference's FastSLAM object
through its own control schedule under a seeded np.random, records every random number a step
consumed and the full particle state after it (tests/golden/fastslam_*.npz);
tests/test_fastslam_oracle_golden.py checks this module against those records bit for bit.

Reference file: 13_fast_slam/fast_slam1.py
"""
import numpy as np

LANDMARKS = np.array([[5, 10, 0], [10, 10, 0], [15, 10, 0], [20, 10, 0],
                      [20, 0, 0], [15, 0, 0], [10, 0, 0], [5, 0, 0]], dtype=np.float32)   # fast_slam1.py:9-20
LANDMARKS[:, 1] -= 11.0
TM = 2
DT = 0.1 / TM                       # :24-25
R_MAX = 10.0                        # :27
STD_R = 0.5                         # :35
STD_PHI = 3 * np.pi / 180.0         # :36
P0 = 0.01                           # :38
THRES_SIG = 2                       # :39
ALPHA = [0.02, 0.01, 0.01, 0.02, 0.01, 0.01]   # :41-43


class Feat:
    __slots__ = ("m", "Cov", "i", "w", "id", "z_pred", "H", "Q", "invQ", "updated")

    def __init__(self, m, Cov):
        self.m, self.Cov, self.i, self.w, self.id = m, Cov, 1, 0, 0
        self.z_pred = self.H = self.Q = self.invQ = None
        self.updated = True

    def copy(self):
        f = Feat(self.m.copy(), self.Cov.copy())
        f.i, f.w, f.id, f.updated = self.i, self.w, self.id, self.updated
        return f


class Part:
    def __init__(self):
        self.x = np.array([[0, 0, 0]]).T     # :106 (integer zeros; the first sampled pose is float64)
        self.feature = []
        self.w = 0
        self.new_feature_id = 0

    def copy(self):
        y = Part()
        y.x = self.x
        y.feature = [f.copy() for f in self.feature]
        y.w, y.new_feature_id = self.w, self.new_feature_id
        return y


def true_pose_step(x_gt, ut, dt=DT):
    """fast_slam1.py:113-122 (no straight-line branch: the driver never passes wt = 0)."""
    vt, wt = ut
    x0, y0, th0 = x_gt
    return [x0 + -vt / wt * np.sin(th0) + vt / wt * np.sin(th0 + wt * dt),
            y0 + vt / wt * np.cos(th0) - vt / wt * np.cos(th0 + wt * dt),
            th0 + wt * dt]


def true_measurements(x_gt, landmarks=LANDMARKS):
    """Noise-free (r, phi) of every landmark from the true pose and whether it is sensed
    (fast_slam1.py:131-142)."""
    x, y, theta = x_gt
    out = []
    for mx, my, _ in landmarks:
        dx = mx - x
        dy = my - y
        r = np.sqrt(dx ** 2 + dy ** 2)
        phi = np.arctan2(dy, dx) - theta
        out.append((r, phi, bool(0 < r <= R_MAX)))
    return out


def noisy_measurement(r, phi, valid, n_r, n_phi):
    """fast_slam1.py:142-157 given the two standard-normal draws of a valid sensing."""
    if valid:
        r = r + n_r * STD_R
        phi = phi + n_phi * STD_PHI
        if phi > np.pi:
            phi -= 2 * np.pi
        elif phi < -np.pi:
            phi += 2 * np.pi
        return np.array([[r, phi]]).T
    return np.array([[R_MAX + 1.0, phi]]).T


def sample_pose(x, ut, ints, dt=DT):
    """fast_slam1.py:160-189: Table 5.3 with the twelve-integer normal sampler (no wt = 0 guard)."""
    vt, wt = ut

    def sample(b, row):
        r = row * 0.001
        return b / 6.0 * r.sum()

    v_hat = vt + sample(ALPHA[0] * abs(vt) + ALPHA[1] * abs(wt), ints[0])
    w_hat = wt + sample(ALPHA[2] * abs(vt) + ALPHA[3] * abs(wt), ints[1])
    g_hat = sample(ALPHA[4] * abs(vt) + ALPHA[5] * abs(wt), ints[2])
    px, py, th = x[0], x[1], x[2]
    x_t = px - v_hat / w_hat * np.sin(th) + v_hat / w_hat * np.sin(th + w_hat * dt)
    y_t = py + v_hat / w_hat * np.cos(th) - v_hat / w_hat * np.cos(th + w_hat * dt)
    th_t = th + w_hat * dt + g_hat * dt
    return np.array([x_t, y_t, th_t]).reshape([3, 1])


def h(m, xt):           # :206-222
    dx = m[0, 0] - xt[0, 0]
    dy = m[1, 0] - xt[1, 0]
    r = np.sqrt(dx ** 2 + dy ** 2)
    phi = np.arctan2(dy, dx) - xt[2, 0]
    if phi > np.pi:
        phi -= np.pi * 2
    elif phi < -np.pi:
        phi += np.pi * 2
    return np.array([[r, phi]]).T


def jac(m, xt):         # :224-238
    dx = m[0, 0] - xt[0, 0]
    dy = m[1, 0] - xt[1, 0]
    q = dx ** 2 + dy ** 2
    r = np.sqrt(q)
    return np.array([[dx / r, dy / r], [-dy / q, dx / q]])


def _wrap_residual(r):
    if r[1, 0] > np.pi:
        r[1, 0] -= np.pi * 2
    elif r[1, 0] < -np.pi:
        r[1, 0] += np.pi * 2
    return r


QT = np.array([[STD_R ** 2, 0.0], [0.0, STD_PHI ** 2]])   # :264-269


def particle_step(y, ut, ints, normals, truth, dt=DT):
    """One particle through fast_slam1.py:246-404.  normals: the standard-normal draws of this
    particle in consumption order (two per sensed landmark)."""
    x1 = sample_pose(y.x[:, 0], ut, ints.astype(np.int64), dt)
    y.x = x1
    for f in y.feature:                                             # :259-282
        f.z_pred = h(f.m, x1)
        f.H = jac(f.m, x1)
        f.Q = np.matmul(np.matmul(f.H, f.Cov), f.H.T) + QT
        f.invQ = np.linalg.inv(f.Q)
        f.updated = False
    n_old = len(y.feature)
    used = 0
    for (r0, phi0, valid) in truth:                                 # :286-375
        if valid:
            z = noisy_measurement(r0, phi0, True, normals[used], normals[used + 1])
            used += 2
        else:
            z = noisy_measurement(r0, phi0, False, 0.0, 0.0)
        if not z[0, 0] <= R_MAX:
            continue
        c, w_max = -1, 0
        for j in range(n_old):
            f = y.feature[j]
            res = _wrap_residual(z - f.z_pred)
            d2 = np.matmul(np.matmul(res.T, f.invQ), res)
            if d2 <= THRES_SIG ** 2:
                eta = 1 / (2 * np.pi * np.sqrt(np.linalg.det(f.Q)))
                w = (eta * np.exp(-0.5 * d2))[0, 0]
                if w_max < w:
                    w_max, c = w, j
        if c == -1:                                                 # :318-347
            th = x1[2, 0]
            m = np.array([[z[0, 0] * np.cos(z[1, 0] + th) + x1[0, 0], z[0, 0] * np.sin(z[1, 0] + th) + x1[1, 0]]]).T
            H = jac(m, x1)
            invH = np.linalg.inv(H)
            f = Feat(m, np.matmul(np.matmul(invH, QT), invH.T))
            f.H, f.updated, f.w, f.id = H, True, P0, y.new_feature_id
            y.new_feature_id += 1
            y.feature.append(f)
        else:                                                       # :348-375
            f = y.feature[c]
            K = np.matmul(np.matmul(f.Cov, f.H.T), f.invQ)
            res = _wrap_residual(z - f.z_pred)
            f.m = f.m + np.matmul(K, res)
            f.Cov = np.matmul((np.eye(2) - np.matmul(K, f.H)), f.Cov)
            f.i += 1
            f.updated, f.w = True, w_max
    y.w = 1.0                                                       # :377-404
    n_assoc = 0
    it = 0
    while it < len(y.feature):      # a Python for-loop over a list that shrinks under it (:380, :401)
        f = y.feature[it]
        it += 1
        if f.updated:
            y.w *= f.w
            n_assoc += 1
        else:
            dx = f.m[0, 0] - x1[0, 0]
            dy = f.m[1, 0] - x1[1, 0]
            f.w = P0
            if dx ** 2 + dy ** 2 <= R_MAX ** 2:
                f.i -= 1
                if f.i < 0:
                    y.feature.remove(f)   # the element that slides into this slot is skipped
    if n_assoc == 0:
        y.w = P0
    return used


def resample_indices(w, r_uniform):
    """fast_slam1.py:411-431: Python-sum normalisation with a 1e-8 floor, then the guarded
    low-variance walk.  Returns (indices, normalised weights)."""
    M = len(w)
    total = 0
    for v in w:
        total += v
    total = max(1e-8, total)
    wn = [v / total for v in w]
    c = wn[0]
    i = 0
    r = r_uniform * 1 / M
    idx = []
    for m in range(M):
        U = r + (m / M)
        while U > c and i < (M - 1):
            i += 1
            c = c + wn[i]
        idx.append(i)
    return np.array(idx), np.array(wn)


class OracleFastSLAM:
    def __init__(self, M=100, landmarks=LANDMARKS):
        self.M, self.landmarks = M, landmarks
        self.x_gt = [0, 0, 0]
        self.Y = [Part() for _ in range(M)]
        self.last_idx = None

    def step(self, ut, ints, normals, r_uniform):
        self.x_gt = true_pose_step(self.x_gt, ut)
        truth = true_measurements(self.x_gt, self.landmarks)
        for k, y in enumerate(self.Y):
            particle_step(y, ut, ints[k], normals[k], truth)
        idx, wn = resample_indices([y.w for y in self.Y], r_uniform)
        for k, y in enumerate(self.Y):
            y.w = wn[k]
        self.Y = [self.Y[i].copy() for i in idx]
        self.last_idx = idx

    def state(self, cap):
        M = self.M
        pose = np.stack([y.x[:, 0] for y in self.Y]).astype(np.float64)
        fm = np.zeros((M, cap, 2)); fc = np.zeros((M, cap, 2, 2)); fi = np.zeros((M, cap), np.int64)
        fw = np.zeros((M, cap)); fid = np.full((M, cap), -1, np.int64)
        for k, y in enumerate(self.Y):
            for j, f in enumerate(y.feature):
                fm[k, j], fc[k, j], fi[k, j], fw[k, j], fid[k, j] = f.m[:, 0], f.Cov, f.i, f.w, f.id
        return {"pose": pose, "w": np.array([y.w for y in self.Y], np.float64),
                "nfeat": np.array([len(y.feature) for y in self.Y]), "new_id": np.array([y.new_feature_id for y in self.Y]),
                "f_m": fm, "f_cov": fc, "f_i": fi, "f_w": fw, "f_id": fid}
