"""mclhip_raycast_static -- the sector query for callers that cannot allocate device memory -- against
mclhip_raycast, through the C ABI: ranges and hit cells bit for bit, on two calls in a row (the second with fewer
poses, so stale contents of the shared device array would show), then after a change of map and sensor, and the
limits."""
import numpy as np
import pytest

from oracle import mcl_oracle as O

pytestmark = pytest.mark.gpu


def _sensor(ctx, angles):
    sp = O.SensorParams()
    ctx.set_sensor(angles, sp.z_max, sp.sigma_hit, sp.lambda_short, [sp.w_hit, sp.w_short, sp.w_max, sp.w_rand],
                   5 * np.pi / 180.0)


def test_static_stage_query_matches_the_allocating_one():
    from slam_amd import _lib as L
    from slam_amd import workload as W
    rng = np.random.RandomState(11)
    grid = W.synthetic_map(256)
    angles = W.beam_angles(16)
    ctx = L.Context(0, L.F64)
    ctx.set_map(grid, W.RES, W.RES)
    _sensor(ctx, angles)
    ext = 256 * W.RES

    def poses(n):
        return np.stack([rng.uniform(-0.5, ext + 0.5, n), rng.uniform(-0.5, ext + 0.5, n), rng.uniform(-7, 7, n)], 1)

    P1, P2 = poses(4096), poses(37)
    for P in (P1, P2):                       # two calls in a row, the second smaller
        R, Hc = ctx.raycast(P)
        Rs, Hs = ctx.raycast(P, static=True)
        assert np.array_equal(R, Rs) and np.array_equal(Hc, Hs)
        assert np.any(R < W.Z_MAX) and np.any(Hc >= 0)
    # the oracle agrees on a subset (the allocating entry point is checked against it at length elsewhere)
    lut, cell = O.obstacle_lut(grid)
    Ro, Ho = O.sector_query_batch(P2, lut, cell, angles)
    Rs, Hs = ctx.raycast(P2, static=True)
    assert np.array_equal(Rs, Ro) and np.array_equal(Hs.astype(np.int64), Ho)
    # without hit cells; one pose (the serving loop's case: the true pose of every step)
    Rs1, none = ctx.raycast(P2[:1], want_hits=False, static=True)
    assert none is None and np.array_equal(Rs1, Ro[:1])
    # a second context on the same GPU uses the same device array, one call after the other
    ctx2 = L.Context(0, L.F64)
    g2 = np.full((40, 40), 255, np.uint8)
    g2[rng.randint(0, 40, 60), rng.randint(0, 40, 60)] = 0
    _sensor(ctx2, [0, 90, 180, 270, 45])     # sensor first, map second
    ctx2.set_map(g2, O.RES_X, O.RES_Y)
    Q = np.stack([rng.uniform(0, 6, 50), rng.uniform(0, 6, 50), rng.uniform(-4, 4, 50)], 1)
    Ra, Ha = ctx2.raycast(Q)
    Rb, Hb = ctx2.raycast(Q, static=True)
    assert np.array_equal(Ra, Rb) and np.array_equal(Ha, Hb)
    Rs, Hs = ctx.raycast(P2, static=True)
    assert np.array_equal(Rs, Ro) and np.array_equal(Hs.astype(np.int64), Ho)
    # limits: more than 4096 poses, more than 65536 pose-beam pairs, nothing set
    with pytest.raises(L.MclHipError, match="at most 4096"):
        ctx.raycast(poses(4097), static=True)
    _sensor(ctx, W.beam_angles(180))
    with pytest.raises(L.MclHipError, match="65536"):
        ctx.raycast(poses(400), static=True)
    R, Hc = ctx.raycast(P2)
    Rs, Hs = ctx.raycast(P2, static=True)    # 37 x 180 fits
    assert np.array_equal(R, Rs) and np.array_equal(Hc, Hs)
    ctx3 = L.Context(0, L.F64)
    with pytest.raises(L.MclHipError, match="no map"):
        ctx3.raycast(P2, static=True)
    for c in (ctx, ctx2, ctx3):
        c.close()
