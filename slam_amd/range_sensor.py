"""Beam range-sensor model with the call surface of the reference's
08_grid_and_montecarlo_localization/range_sensor.py (class RangeSensor,
Eq. 6.13 of Probabilistic Robotics); the mixture is evaluated by the
`mclhip_compute_px` kernel, not on the host.
"""
import numpy as np

from . import _lib


class RangeSensor:
    """Same attributes and methods as range_sensor.py:4-126.

    Attributes may be changed after construction (the reference overrides
    `z_max` that way, mc_localization.py:56); they are pushed to the device
    again whenever they differ from what the device last received.
    """

    def __init__(self, device=0):
        # range_sensor.py:5-28
        self.sigma_hit = 0.15
        self.lambda_short = 1.0
        self.z_max = 5
        w_hit, w_short, w_max, w_rand = 0.97, 0.005, 0.020, 0.005
        total = w_hit + w_short + w_max + w_rand
        self.w_hit = w_hit / total
        self.w_short = w_short / total
        self.w_max = w_max / total
        self.w_rand = w_rand / total
        self.resolution = 0.01
        self.pdf = np.zeros(self.z_max)
        self._device = device
        self._ctx = None
        self._pushed = None

    # -- device plumbing ----------------------------------------------------
    def constants(self):
        return (float(self.z_max), float(self.sigma_hit), float(self.lambda_short),
                (float(self.w_hit), float(self.w_short), float(self.w_max), float(self.w_rand)))

    def _eval(self, z, z_true, w4=None):
        if self._ctx is None:
            self._ctx = _lib.Context(self._device, _lib.F64)
        z_max, sigma, lam, w = self.constants()
        w = w if w4 is None else w4
        sig = (z_max, sigma, lam, w)
        if sig != self._pushed:
            self._ctx.set_sensor([0.0], z_max, sigma, lam, w, 5 * np.pi / 180.0)
            self._pushed = sig
        return self._ctx.compute_px(z, z_true)

    def _scalar(self, z, z_true, w4=None):
        return np.float64(self._eval(np.array([z], np.float64), np.array([z_true], np.float64), w4)[0])

    # -- reference surface ----------------------------------------------------
    def compute_px(self, z, z_true):
        """p(z | z*), range_sensor.py:95-112.  Scalars or arrays (broadcast)."""
        if np.ndim(z) == 0 and np.ndim(z_true) == 0:
            return self._scalar(z, z_true)
        return self._eval(z, z_true)

    # single components = the mixture with a one-hot weight vector (x*1 + 0 + 0 + 0 is exact)
    def compute_p_hit(self, x, z):
        return self._scalar(x, z, (1.0, 0.0, 0.0, 0.0))

    def compute_p_short(self, x, z):
        return self._scalar(x, z, (0.0, 1.0, 0.0, 0.0))

    def compute_p_max(self, x):
        return self._scalar(x, 1.0, (0.0, 0.0, 1.0, 0.0))

    def compute_p_rand(self, x):
        return self._scalar(x, 1.0, (0.0, 0.0, 0.0, 1.0))

    def compute_pdf(self, z):
        """Mixture over the lattice x = i*resolution, i = 0..z_max/resolution (range_sensor.py:69-93)."""
        n = int(self.z_max / self.resolution) + 1
        x = np.arange(n) * self.resolution
        self.pdf = self._eval(x, np.full(n, float(z)))
        return self.pdf

    def measure(self):
        """Draw a range from the tabulated pdf (range_sensor.py:114-126): first lattice point whose
        running sum reaches a uniform draw scaled by the total."""
        r = np.random.random() * self.pdf.sum()
        acc = 0
        for i, p in enumerate(self.pdf):
            if acc >= r:
                return i * self.resolution
            acc += p
        return 0
