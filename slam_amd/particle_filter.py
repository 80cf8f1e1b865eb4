"""One-dimensional particle filter with the call surface of the reference's
04_nonparametric_filters/particle_filter.py (class ParticleFilter; the script's plotting loop is
not part of it).

`resample` -- the low-variance sampler of Table 4.4 (particle_filter.py:58-81) -- runs on the GPU
through libmclhip.so: the weights are divided by their sum accumulated in np.sum's order and the
strictly sequential running sum of the walk is reproduced exactly, so the drawn particles are the
ones the reference's loop draws.  The uniform number comes from np.random, as in the reference.
Where the reference's unguarded `while U > c` would run past the last particle (IndexError), the
index is clamped to M - 1.

The proposal draws (`sample`, `sample_uniform`, `sample_gaussian`) consume np.random on the host in
the reference's order; `update_pdf` is the 50-bin histogram of the script.

There is no CPU fallback for the sampler: without the library or a GPU the constructor raises.
"""
import numpy as np

from . import _lib

x_max = 50          # particle_filter.py:97-99
x_obj = 30
x_robot = 0


def normal_dist(mean_x, std_x, x):
    """Normal density (particle_filter.py:5-7)."""
    return 1 / (std_x * np.sqrt(2 * np.pi)) * np.exp(-1 / 2 * ((x - mean_x) / std_x) ** 2)


class ParticleFilter:
    """particle_filter.py:9-95.  The reference's methods read the script's globals `M` and `x_max`;
    here they are the constructor's `M` and the keyword `x_max`."""

    def __init__(self, M, px, x_max=x_max, device=0):
        self.M = int(M)
        self.px = px
        self.w = np.ones(self.M)
        self.X = []
        self.x_max = int(x_max)
        self.last_idx = None
        self._ctx = _lib.Context(device, _lib.F64)

    # -- proposals: host draws, the reference's consumption of np.random ----------------------
    def sample(self):
        """Draw every particle from the tabulated distribution px (:16-35)."""
        self.X = []
        for _ in range(self.M):
            p = np.random.rand()
            cdf = 0
            for i in range(len(self.px)):
                cdf = self.px[i] if i == 0 else cdf + self.px[i]
                if cdf >= p:
                    x = i
                    break
            self.X.append(x)

    def sample_uniform(self):
        """:37-42."""
        self.X = [np.random.randint(low=0, high=self.x_max) for _ in range(self.M)]

    def sample_gaussian(self):
        """:44-56: every particle diffuses with standard deviation 0.5."""
        self.X = [np.random.randn() * 0.5 + self.X[m] for m in range(self.M)]

    # -- Table 4.4 on the GPU -------------------------------------------------------------------
    def resample(self):
        """Low-variance sampler (:58-81)."""
        M = self.M
        r = np.random.rand() * 1 / M
        P = np.zeros((M, 3))
        P[:, 0] = np.asarray(self.X, np.float64)
        self._ctx.set_particles(P, np.asarray(self.w, np.float64))
        self._ctx.normalise(_lib.W_PRODUCT)                 # w / np.sum(w), numpy's summation order
        self.last_idx = self._ctx.resample(r)
        Xn, wn, _ = self._ctx.get_particles()
        self.X = list(Xn[:, 0])
        self.w = list(wn)
        self.update_pdf()

    def update_pdf(self):
        """:83-95: weights accumulated per rounded position, particle by particle, then normalised."""
        if not isinstance(self.px, np.ndarray):
            self.px = np.array(self.px, np.float64)
        self.px[:] = 0.0
        pos = np.clip(np.asarray(self.X, np.float64), 0, self.x_max - 1)
        cells = (pos + 0.5).astype(np.int64)               # int() truncates; the operand is >= 0.5
        np.add.at(self.px, cells, np.asarray(self.w, np.float64))   # unbuffered: adds in particle order
        self.px = self.px / np.sum(self.px)
