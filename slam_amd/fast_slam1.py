"""FastSLAM 1.0 with unknown correspondence with the call surface of the reference's
13_fast_slam/fast_slam1.py (class FastSLAM, module constants, Particle / Feature views).

The filter step (`fast_slam`) runs on the GPU through libmclhip.so (`mclhip_fs_*`).  The host
keeps what concerns the one true pose (`control`, the noise-free part of `measure`) and draws the
random numbers from np.random in the reference's own order -- per particle: three times twelve
integers for the pose sampler, then two standard normals per sensed landmark; finally one
uniform number for the resampler -- so a seeded run consumes the stream exactly as the reference
does.  `noise="device"` switches to the on-device generators for large particle counts.

There is no CPU fallback: without the library or a GPU the constructor raises.
"""
import numpy as np

from . import _lib

landmarks = np.array([[5, 10, 0], [10, 10, 0], [15, 10, 0], [20, 10, 0],
                      [20, 0, 0], [15, 0, 0], [10, 0, 0], [5, 0, 0]], dtype=np.float32)   # fast_slam1.py:9-20
landmarks[:, 1] -= 11.0
L = len(landmarks)

tm = 2
dt = 0.1 / tm
r_max = 10.0
DIM_POSE = 3
DIM_MEAS = 2


def deg2rad(x):
    return x * np.pi / 180.0


STD_R = 0.5
STD_PHI = deg2rad(3)
P0 = 0.01
thres_sig = 2
alpha = [0.02, 0.01, 0.01, 0.02, 0.01, 0.01]


class Feature:
    """Read-only view of one landmark estimate (fast_slam1.py:46-58)."""

    def __init__(self, m=None, Cov=None):
        self.m, self.Cov, self.i, self.w, self.id = m, Cov, 1, 0, 0

    def copy(self):
        """fast_slam1.py:60-77 (the per-step prediction fields are not part of the view)."""
        f = Feature(self.m.copy(), self.Cov.copy())
        f.i, f.w, f.id = self.i, self.w, self.id
        return f


class Particle:
    """Read-only view of one particle (fast_slam1.py:79-84): path, feature list, weight."""

    def __init__(self):
        self.x, self.feature, self.w, self.new_feature_id = [], [], 0, 0

    def copy(self):
        """fast_slam1.py:86-95."""
        y = Particle()
        y.x = self.x.copy()
        y.feature = [f.copy() for f in self.feature]
        y.w, y.new_feature_id = self.w, self.new_feature_id
        return y


class FastSLAM:
    """Same constructor, attributes and methods as fast_slam1.py:97-435 (plotting aside).

    Extra keyword arguments: `M` (particles, 100 in the reference), `landmarks` ([L,3] array, the
    module's eight by default), `capacity` (feature slots per particle), `noise` ("numpy": the
    reference's np.random consumption; "device": on-device generators), `device`.
    """

    def __init__(self, M=100, landmarks=None, capacity=None, noise="numpy", device=0):
        self.x_gt = [[0, 0, 0]]
        self.x = [[0, 0, 0]]
        self.M = int(M)
        self.landmarks = globals()["landmarks"] if landmarks is None else np.asarray(landmarks, np.float32)
        self.noise = noise
        cap = int(capacity) if capacity is not None else 2 * len(self.landmarks) + 8
        self._ctx = _lib.Context(device, _lib.F64)
        self._ctx.fs_init(self.M, cap)
        self._ctx.fs_set_model(STD_R, STD_PHI, r_max, P0, thres_sig)
        x0 = np.array([[0, 0, 0]]).T
        self._paths = [[x0] for _ in range(self.M)]     # Particle.x of every particle (:106-107)
        self._step = 0
        self._Y = None

    # -- the reference's helper methods (host, one pose) ----------------------------
    def control(self, ut):
        """fast_slam1.py:110-124."""
        vt, wt = ut
        x0, y0, theta0 = self.x_gt[-1]
        x_gt = x0 + -vt / wt * np.sin(theta0) + vt / wt * np.sin(theta0 + wt * dt)
        y_gt = y0 + vt / wt * np.cos(theta0) - vt / wt * np.cos(theta0 + wt * dt)
        theta_gt = theta0 + wt * dt
        self.x_gt.append([x_gt, y_gt, theta_gt])
        self.fast_slam(ut)

    def _true_measurement(self, mi):
        mx, my, _ = mi
        x, y, theta = self.x_gt[-1]
        dx = mx - x
        dy = my - y
        r = np.sqrt(dx ** 2 + dy ** 2)
        return r, np.arctan2(dy, dx) - theta

    def measure(self, mi):
        """One noisy (r, phi) of landmark mi from the true pose (fast_slam1.py:127-157)."""
        r, phi = self._true_measurement(mi)
        if 0 < r <= r_max:
            r += np.random.randn() * STD_R
            phi += np.random.randn() * STD_PHI
            if phi > np.pi:
                phi -= 2 * np.pi
            elif phi < -np.pi:
                phi += 2 * np.pi
            return np.array([[r, phi]]).T
        return np.array([[r_max + 1.0, phi]]).T

    def sample_normal_distribution(self, b):
        r = np.random.randint(-1000, 1000, size=12) * 0.001
        return b / 6.0 * r.sum()

    def sample_motion_model_velocity(self, xt_1, ut, dt=0.1):
        """Table 5.3 for one pose, no straight-line branch (fast_slam1.py:167-189): the three times twelve
        integers come from np.random.randint as `sample_normal_distribution` draws them, the pose from the GPU."""
        ints = np.stack([np.random.randint(-1000, 1000, size=12) for _ in range(3)])[None]
        pose = np.array([np.ravel(v)[0] for v in (xt_1[0], xt_1[1], xt_1[2])], np.float64).reshape(1, 3)
        out = self._ctx.motion_book(pose, ut[0], ut[1], dt, alpha, ints)[0]
        shape = np.shape(xt_1[0])
        return [np.reshape(v, shape) if shape else v for v in out]

    def inv_h(self, z, xt):
        return [z[0, 0] * np.cos(z[1, 0] + xt[2, 0]) + xt[0, 0], z[0, 0] * np.sin(z[1, 0] + xt[2, 0]) + xt[1, 0]]

    def h(self, m_t, xt):
        dx = m_t[0, 0] - xt[0, 0]
        dy = m_t[1, 0] - xt[1, 0]
        phi = np.arctan2(dy, dx) - xt[2, 0]
        if phi > np.pi:
            phi -= np.pi * 2
        elif phi < -np.pi:
            phi += np.pi * 2
        return np.array([[np.sqrt(dx ** 2 + dy ** 2), phi]]).T

    def jacobian_H(self, m_t, xt):
        dx = m_t[0, 0] - xt[0, 0]
        dy = m_t[1, 0] - xt[1, 0]
        q = dx ** 2 + dy ** 2
        r = np.sqrt(q)
        return np.array([[dx / r, dy / r], [-dy / q, dx / q]])

    # -- the filter step ---------------------------------------------------------------
    def fast_slam(self, ut):
        """fast_slam1.py:240-433 on the GPU."""
        M = self.M
        if self.noise == "numpy":
            truth = [self._true_measurement(lm) for lm in self.landmarks]       # the reference's scalar arithmetic
            tr = np.array([float(t[0]) for t in truth])
            tp = np.array([float(t[1]) for t in truth])
        else:
            # device generators: no stream to stay in step with, so the landmarks are done in one go (array
            # arithmetic squares by multiplying where the scalar `**` goes through pow: an ulp in rare cases)
            x, y, theta = (float(v) for v in self.x_gt[-1])
            dx = self.landmarks[:, 0].astype(np.float64) - x
            dy = self.landmarks[:, 1].astype(np.float64) - y
            tr = np.sqrt(dx * dx + dy * dy)
            tp = np.arctan2(dy, dx) - theta
        tv = ((0 < tr) & (tr <= r_max)).astype(np.uint8)
        nv = int(tv.sum())
        ints = normals = None
        if self.noise == "numpy":
            ints = np.empty((M, 3, 12), np.int16)
            normals = np.empty((M, 2 * nv))
            for k in range(M):                     # the reference's draw order, particle by particle
                for a in range(3):
                    ints[k, a] = np.random.randint(-1000, 1000, size=12)
                for j in range(2 * nv):
                    normals[k, j] = np.random.randn()
        r = np.random.rand() * 1 / M
        self._step += 1
        idx = self._ctx.fs_step(ut[0], ut[1], dt, alpha, tr, tp, tv, ints, normals, r, seed=self._step,
                                want_idx=M <= 100000)
        self._Y = None
        if idx is not None and self._paths is not None:
            pose = self._ctx.fs_get_state()["pose"] if M <= 4096 else None
            if pose is None:
                self._paths = None                 # paths are a plotting aid; not kept for large clouds
            else:
                self._paths = [self._paths[i] + [pose[k].reshape(3, 1)] for k, i in enumerate(idx)]
        else:
            self._paths = None

    @property
    def Y(self):
        """The particle list as objects (built on demand from the device state)."""
        if self._Y is None:
            st = self._ctx.fs_get_state()
            out = []
            for k in range(self.M):
                y = Particle()
                y.x = self._paths[k] if self._paths is not None else [st["pose"][k].reshape(3, 1)]
                y.w = st["w"][k]
                y.new_feature_id = int(st["new_id"][k])
                for j in range(int(st["nfeat"][k])):
                    f = Feature(st["f_m"][k, j].reshape(2, 1).copy(), st["f_cov"][k, j].copy())
                    f.i, f.w, f.id = int(st["f_i"][k, j]), st["f_w"][k, j], int(st["f_id"][k, j])
                    y.feature.append(f)
                out.append(y)
            self._Y = out
        return self._Y

    def state(self):
        """Raw arrays of the device state (poses, weights, feature slots)."""
        return self._ctx.fs_get_state()
