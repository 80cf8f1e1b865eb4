"""Synthetic inputs of the benchmark configurations (maps, beam sets, particle clouds).

Everything is generated from fixed numpy RandomState seeds so that the GPU path,
the CPU checker and every rank of a multi-GPU run see identical inputs.
"""
import numpy as np

RES = 0.15          # cell size in metres (mc_localization.py:9-10)
Z_MAX = 5           # mc_localization.py:14
SEED_MAP, SEED_PARTICLES, SEED_STREAM = 1234, 1235, 1236


def synthetic_map(n, seed=SEED_MAP, target_occupancy=0.04):
    """uint8 [n,n] with values {0,255}: a 2-cell border wall plus random axis-aligned wall
    segments (length U{8..64} cells, thickness U{1,2}), added in batches of n*n/16384 until the
    occupied fraction reaches `target_occupancy`."""
    rng = np.random.RandomState(seed)
    m = np.full((n, n), 255, np.uint8)
    m[:2, :] = 0
    m[-2:, :] = 0
    m[:, :2] = 0
    m[:, -2:] = 0
    batch = max(1, n * n // 16384)
    while True:
        length = rng.randint(8, 65, batch)
        thick = rng.randint(1, 3, batch)
        i0 = rng.randint(0, n, batch)
        j0 = rng.randint(0, n, batch)
        horiz = rng.randint(0, 2, batch)
        for s in range(batch):
            if horiz[s]:
                m[i0[s]:i0[s] + thick[s], j0[s]:j0[s] + length[s]] = 0
            else:
                m[i0[s]:i0[s] + length[s], j0[s]:j0[s] + thick[s]] = 0
        if np.count_nonzero(m == 0) >= target_occupancy * n * n:
            return m


def beam_angles(B):
    """Evenly spaced beams over one turn, in degrees."""
    return [k * 360.0 / B for k in range(B)]


def uniform_particles(grid, M, seed=SEED_PARTICLES, res=RES):
    """M poses uniform over the free cells (+- half a cell of jitter), heading U[0, 2pi)."""
    rng = np.random.RandomState(seed)
    H, W = grid.shape
    X = np.zeros((M, 3))
    filled = 0
    while filled < M:
        need = M - filled
        i = rng.randint(0, H, int(need * 1.2) + 16)
        j = rng.randint(0, W, len(i))
        ok = grid[i, j] == 255
        i, j = i[ok][:need], j[ok][:need]
        k = len(i)
        X[filled:filled + k, 0] = j * res
        X[filled:filled + k, 1] = i * res
        filled += k
    X[:, 0] += rng.uniform(-res / 2, res / 2, M)
    X[:, 1] += rng.uniform(-res / 2, res / 2, M)
    X[:, 2] = rng.uniform(0, 2 * np.pi, M)
    return X


def controls(n_steps):
    """Alternating straight / turning controls (mc_localization.py:324-326 uses the first)."""
    return [(1.5, 0.0) if t % 2 == 0 else (1.5, 0.5) for t in range(n_steps)]
