"""mclhip_raycast_static -- the sector query for callers that cannot allocate device memory -- against
mclhip_raycast, through the C ABI: ranges and hit cells bit for bit, on two calls in a row (the second with fewer
poses, so stale contents of the shared device array would show), then after a change of map and sensor, and the
limits; that the call leaves the device's free memory as it found it; and two contexts driven from two threads."""
import ctypes
import threading

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


def _free_device_bytes():
    """hipMemGetInfo of the current device, asked of a HIP runtime this process has already loaded and initialised: the
    one the library is linked against (under /opt/rocm) first.  A process that has imported torch as well holds torch's
    own copy of the runtime too, and a bare dlopen("libamdhip64.so") may then come back with an instance that sees no
    device."""
    paths = {line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line}
    assert paths, "no HIP runtime loaded"
    codes = []
    for path in sorted(paths, key=lambda q: (not q.startswith("/opt/rocm"), q)):
        hip = ctypes.CDLL(path)
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        rc = hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
        if rc == 0:
            return free.value
        codes.append((path, rc))
    raise AssertionError("hipMemGetInfo failed: %r" % (codes,))


def test_static_stage_query_allocates_no_device_memory():
    from slam_amd import _lib as L
    from slam_amd import workload as W
    rng = np.random.RandomState(3)
    grid = W.synthetic_map(256)
    ctx = L.Context(0, L.F64)
    ctx.set_map(grid, W.RES, W.RES)
    _sensor(ctx, W.beam_angles(16))
    P = np.stack([rng.uniform(0, 38, 4096), rng.uniform(0, 38, 4096), rng.uniform(-3, 3, 4096)], 1)
    ctx.raycast(P, static=True)                       # first call: whatever the runtime sets up lazily happens here
    ctx.sync()
    import gc
    gc.collect()                                      # (contexts of earlier tests give their memory back now, not in between)
    before = _free_device_bytes()
    low = before
    for n in (4096, 1, 300):
        ctx.raycast(P[:n], static=True)
        low = min(low, _free_device_bytes())
    assert low == before, (before, low)
    ctx.raycast(P)                                    # the allocating entry point, for contrast, frees what it took
    assert _free_device_bytes() == before
    ctx.close()


def test_two_threads_with_a_context_each_share_the_static_array():
    """Two contexts on one GPU, each driven by its own thread: the calls take turns on the device's array (a mutex per
    device inside the library), so neither sees the other's poses or ranges."""
    from slam_amd import _lib as L
    from slam_amd import workload as W
    rng = np.random.RandomState(17)
    jobs = []
    for k, (side, beams, n) in enumerate(((256, 16, 2000), (128, 45, 1400))):
        grid = W.synthetic_map(side, seed=40 + k)
        ctx = L.Context(0, L.F64)
        ctx.set_map(grid, W.RES, W.RES)
        _sensor(ctx, W.beam_angles(beams))
        ext = side * W.RES
        P = np.stack([rng.uniform(0, ext, n), rng.uniform(0, ext, n), rng.uniform(-3, 3, n)], 1)
        jobs.append((ctx, P, ctx.raycast(P)))
    bad, errors = [0, 0], []

    def work(k):
        ctx, P, (R, Hc) = jobs[k]
        try:
            for _ in range(40):
                Rs, Hs = ctx.raycast(P, static=True)
                if not (np.array_equal(R, Rs) and np.array_equal(Hc, Hs)):
                    bad[k] += 1
        except Exception as e:           # noqa: BLE001 -- reported by the assertion below
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert bad == [0, 0], bad
    for ctx, _, _ in jobs:
        ctx.close()
