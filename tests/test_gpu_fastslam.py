"""FastSLAM 1.0 on the GPU (through the C ABI) against the state the reference's FastSLAM object
reached under the same random numbers (tests/golden/fastslam_*.npz).  Needs an MI355X: `-m gpu`.

Bars: ancestor indices, feature counts, feature ids and observation counters bit-exact at every
step; poses, weights, feature means within 1e-12 relative; feature covariances within 1e-11 of
the matrix's largest entry (the EKF update forms (I - K H) Cov, whose small entries are
differences of products: an ulp of numpy's BLAS-ordered 2x2 products is not an ulp of the result).
"""
import numpy as np
import pytest

from oracle import fastslam_oracle as F
from util import load, rel_err, steps

pytestmark = pytest.mark.gpu


def _truth(x_gt, landmarks):
    t = F.true_measurements(list(x_gt), landmarks)
    return np.array([float(a[0]) for a in t]), np.array([float(a[1]) for a in t]), np.array([a[2] for a in t], np.uint8)


def _compare(st, s, t):
    n = s["f_m"].shape[1]
    assert np.array_equal(st["nfeat"], s["nfeat"]), t
    assert np.array_equal(st["new_id"], s["new_id"]), t
    assert np.array_equal(st["f_id"][:, :n], s["f_id"]), t
    assert np.array_equal(st["f_i"][:, :n], s["f_i"]), t
    assert rel_err(st["pose"], s["pose"]) < 1e-12, (t, rel_err(st["pose"], s["pose"]))
    assert rel_err(st["w"], s["w"]) < 1e-12, (t, rel_err(st["w"], s["w"]))
    assert rel_err(st["f_w"][:, :n], s["f_w"]) < 1e-11, (t, rel_err(st["f_w"][:, :n], s["f_w"]))
    assert rel_err(st["f_m"][:, :n], s["f_m"]) < 1e-12, (t, rel_err(st["f_m"][:, :n], s["f_m"]))
    scale = np.max(np.abs(s["f_cov"]), axis=(2, 3), keepdims=True)
    err = np.abs(st["f_cov"][:, :n] - s["f_cov"])
    assert np.all(err <= 1e-11 * scale), (t, float(np.max(err / np.maximum(scale, 1e-300))))


@pytest.mark.parametrize("name", ["fastslam_seed0", "fastslam_seed3"])
def test_follows_the_reference_step_by_step(name):
    from slam_amd import _lib as L
    g = load(name)
    M = int(g["M"])
    lm = g["landmarks"].astype(np.float32)
    ctx = L.Context(0, L.F64)
    ctx.fs_init(M, 16)
    # independent chain: the device carries its own state from step to step
    x_gt = [0, 0, 0]
    for t, s in enumerate(steps(g)):
        ut = [float(s["ut"][0]), float(s["ut"][1])]
        x_gt = F.true_pose_step(x_gt, ut)
        tr, tp, tv = _truth(x_gt, lm)
        r = float(s["r_uniform"][0]) * 1 / M
        idx = ctx.fs_step(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, s["ints"], s["normals"], r, want_idx=True)
        _compare(ctx.fs_get_state(), s, t)


def test_ancestors_equal_the_oracle_walk():
    from slam_amd import _lib as L
    g = load("fastslam_seed0")
    M = int(g["M"])
    lm = g["landmarks"].astype(np.float32)
    ctx = L.Context(0, L.F64)
    ctx.fs_init(M, 16)
    orc = F.OracleFastSLAM(M)
    x_gt = [0, 0, 0]
    for t, s in enumerate(steps(g)[:12]):
        ut = [float(s["ut"][0]), float(s["ut"][1])]
        x_gt = F.true_pose_step(x_gt, ut)
        tr, tp, tv = _truth(x_gt, lm)
        idx = ctx.fs_step(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, s["ints"], s["normals"],
                          float(s["r_uniform"][0]) * 1 / M, want_idx=True)
        orc.step(ut, s["ints"], s["normals"], float(s["r_uniform"][0]))
        assert np.array_equal(idx, orc.last_idx), t


def test_capacity_overflow_is_reported():
    from slam_amd import _lib as L
    g = load("fastslam_seed0")
    M = int(g["M"])
    lm = g["landmarks"].astype(np.float32)
    ctx = L.Context(0, L.F64)
    ctx.fs_init(M, 1)
    s = steps(g)[0]
    ut = [float(s["ut"][0]), float(s["ut"][1])]
    tr, tp, tv = _truth(F.true_pose_step([0, 0, 0], ut), lm)
    with pytest.raises(L.MclHipError, match="feature slots"):
        ctx.fs_step(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, s["ints"], s["normals"], 0.001)


def test_large_population_with_device_noise():
    """2^18 particles, 64 landmarks on a ring, device generators: properties only."""
    from slam_amd import _lib as L
    M, nl = 1 << 18, 64
    ang = np.arange(nl) * 2 * np.pi / nl
    lm = np.stack([12 * np.cos(ang) + 3, 12 * np.sin(ang) - 6, np.zeros(nl)], 1).astype(np.float32)
    ctx = L.Context(0, L.F64)
    ctx.fs_init(M, 96)
    x_gt = [0, 0, 0]
    for t in range(6):
        ut = [25, 0.001] if t < 4 else [25, -np.pi / 2 * 5]
        x_gt = F.true_pose_step(x_gt, ut)
        tr, tp, tv = _truth(x_gt, lm)
        idx = ctx.fs_step(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, None, None, 0.37 / M, seed=100 + t, want_idx=True)
        assert np.all(np.diff(idx) >= 0) and idx[0] >= 0 and idx[-1] < M
    st = ctx.fs_get_state()
    assert np.all(np.isfinite(st["pose"])) and np.all(st["nfeat"] > 0) and np.all(st["nfeat"] <= 96)
    assert np.all(np.isfinite(st["w"])) and np.all(st["w"] > 0) and np.all(st["w"] <= 1)
    # pose cloud centred near the true pose
    assert np.linalg.norm(st["pose"][:, :2].mean(0) - np.array(x_gt[:2], float)) < 1.0
    det = st["f_cov"][:, :, 0, 0] * st["f_cov"][:, :, 1, 1] - st["f_cov"][:, :, 0, 1] * st["f_cov"][:, :, 1, 0]
    live = np.arange(96)[None, :] < st["nfeat"][:, None]
    assert np.all(det[live] > 0)


def test_dropin_class_follows_reference_run():
    """slam_amd.fast_slam1.FastSLAM driven like fast_slam1.py:536-561 under the same np.random seed:
    the host draws the random numbers in the reference's order, so every step sees the same stream."""
    from slam_amd import fast_slam1 as S
    g = load("fastslam_seed3")
    np.random.seed(int(g["seed"]))
    slam = S.FastSLAM()
    for t, s in enumerate(steps(g)):
        ut = [float(s["ut"][0]), float(s["ut"][1])]
        slam.control(ut)
        assert np.array_equal(np.array(slam.x_gt[-1], np.float64), s["x_gt"])
        _compare(slam.state(), s, t)
    Y = slam.Y
    assert len(Y) == slam.M and len(Y[0].x) == len(steps(g)) + 1
    assert [len(y.feature) for y in Y] == list(s["nfeat"])
    assert Y[5].feature[0].m.shape == (2, 1) and Y[5].feature[0].Cov.shape == (2, 2)


def test_two_measurements_that_pick_the_same_feature():
    """Landmarks in tight pairs: both measurements of a pair often associate with one feature, which
    is then updated twice in one step -- the second time with the first update's mean and covariance
    but the step's original prediction (fast_slam1.py:348-375 reads z_pred, H, invQ of :259-282)."""
    from slam_amd import _lib as L
    M, cap = 96, 24
    base = np.array([[4.0, 1.0], [6.0, -2.0], [3.0, -3.0], [7.0, 2.5]])
    lm = np.concatenate([base, base + [0.25, 0.1], base[:2] + [-0.1, 0.3]])
    lm = np.concatenate([lm, np.zeros((len(lm), 1))], 1).astype(np.float32)
    ctx = L.Context(0, L.F64)
    ctx.fs_init(M, cap)
    orc = F.OracleFastSLAM(M, lm)
    rng = np.random.RandomState(21)
    x_gt = [0, 0, 0]
    doubles = 0
    prev_i = None
    for t in range(8):
        ut = [8.0, 0.3] if t % 3 else [8.0, -0.4]
        x_gt = F.true_pose_step(x_gt, ut)
        tr, tp, tv = _truth(x_gt, lm)
        nv = int(tv.sum())
        ints = rng.randint(-1000, 1000, size=(M, 3, 12)).astype(np.int16)
        normals = rng.randn(M, 2 * nv)
        ru = float(rng.rand())
        idx = ctx.fs_step(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, ints, normals, ru / M, want_idx=True)
        orc.step(ut, ints, normals, ru)
        assert np.array_equal(idx, orc.last_idx), t
        s = orc.state(cap)
        _compare(ctx.fs_get_state(), s, t)
        if prev_i is not None:
            doubles += int(np.sum(s["f_i"] - prev_i[idx] >= 2))
        prev_i = s["f_i"]
    assert doubles > 20, doubles     # the case this test is about did occur


def test_dense_landmark_field_against_the_oracle():
    """256 landmarks, about a quarter of them in range at a time (the layout of the large timing
    configuration): the association sweep's landmark index by bearing and its range pre-test see
    windows that wrap around the circle and features far out of range."""
    from slam_amd import _lib as L
    M, nl, cap = 128, 256, 160
    rng = np.random.RandomState(9)
    lm = np.stack([rng.uniform(-8, 33, nl), rng.uniform(-24, 8, nl), np.zeros(nl)], 1).astype(np.float32)
    ctx = L.Context(0, L.F64)
    ctx.fs_init(M, cap)
    ctx2 = L.Context(0, L.F64)
    ctx2.fs_init(M, cap)
    orc = F.OracleFastSLAM(M, lm)
    x_gt = [0, 0, 0]
    one_ancestor_tiles = mixed_tiles = 0
    for t in range(7):
        ut = [25, 0.001] if t % 5 else [25, -np.pi / 2 * 5]
        x_gt = F.true_pose_step(x_gt, ut)
        tr, tp, tv = _truth(x_gt, lm)
        nv = int(tv.sum())
        ints = rng.randint(-1000, 1000, size=(M, 3, 12)).astype(np.int16)
        normals = rng.randn(M, 2 * nv)
        ru = float(rng.rand())
        idx = ctx.fs_step(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, ints, normals, ru / M, want_idx=True)
        idx2 = ctx2.fs_step(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, ints, normals, ru / M, want_idx=True)
        orc.step(ut, ints, normals, ru)
        assert np.array_equal(idx, orc.last_idx) and np.array_equal(idx2, orc.last_idx), t
        _compare(ctx.fs_get_state(), orc.state(cap), t)
        one_ancestor_tiles += sum(int(len(set(idx[k:k + 64].tolist())) == 1) for k in (0, 64))
        mixed_tiles += sum(int(len(set(idx[k:k + 64].tolist())) > 1) for k in (0, 64))
    assert nv > 40 and int(ctx.fs_get_state()["nfeat"].max()) > nv
    # with this many landmarks the weights are peaked: whole tiles of 64 particles descend from one ancestor and keep
    # one copy of its feature slots.  ctx read its state after every step (which gives every particle its own copy
    # again); ctx2 ran all steps on the shared copies and is read only now
    assert one_ancestor_tiles >= 4 and mixed_tiles >= 1, (one_ancestor_tiles, mixed_tiles)
    _compare(ctx2.fs_get_state(), orc.state(cap), 99)
    ctx2.close()


def test_device_measurement_noise_is_standard_normal():
    """The on-device generator of the measurement noise (float32 Box-Muller on a counter-based generator) is its
    own stream: its contract is the distribution.  After one step from the origin every feature is a birth at
    pose + z, so the drawn (range, bearing) noise can be read back from the state: moments, independence of the
    two components and of neighbouring landmarks, and a Kolmogorov-Smirnov test."""
    from scipy import stats
    from slam_amd import _lib as L
    M, nl = 1 << 18, 4      # few landmarks: the weight P0 ** nl of a step of births must stay above the 1e-8 floor of the sum
    ang = np.arange(nl) * 2 * np.pi / nl + 0.3
    lm = np.stack([6 * np.cos(ang) + 1, 6 * np.sin(ang), np.zeros(nl)], 1).astype(np.float32)
    ctx = L.Context(0, L.F64)
    ctx.fs_init(M, nl + 4)
    ut = [25, 0.001]
    x_gt = F.true_pose_step([0, 0, 0], ut)
    tr, tp, tv = _truth(x_gt, lm)
    assert tv.all()
    idx = ctx.fs_step(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, None, None, 0.5 / M, seed=77, want_idx=True)
    assert np.array_equal(idx, np.arange(M))            # equal weights: every particle once
    st = ctx.fs_get_state()
    assert np.all(st["nfeat"] == nl)
    d = st["f_m"][:, :nl, :] - st["pose"][:, None, :2]
    zr = np.hypot(d[..., 0], d[..., 1])
    zphi = np.arctan2(d[..., 1], d[..., 0]) - st["pose"][:, None, 2]
    n_r = (zr - tr[None, :]) / F.STD_R
    n_phi = (zphi - tp[None, :] + np.pi) % (2 * np.pi) - np.pi
    n_phi = n_phi / F.STD_PHI
    n = n_r.size
    for v in (n_r, n_phi):
        assert abs(v.mean()) < 5 / np.sqrt(n) and abs(v.var() - 1) < 0.01
        assert abs(stats.skew(v.ravel())) < 0.012 and abs(stats.kurtosis(v.ravel())) < 0.025
        assert stats.kstest(v.ravel()[::37], "norm").pvalue > 1e-3
        assert np.abs(v).max() < 6.7                    # sqrt(-2 ln 2^-25)
    assert abs(np.corrcoef(n_r.ravel(), n_phi.ravel())[0, 1]) < 5 / np.sqrt(n)
    assert abs(np.corrcoef(n_r[:, 0::2].ravel(), n_r[:, 1::2].ravel())[0, 1]) < 5 / np.sqrt(n / 2)   # same generator block
    assert abs(np.corrcoef(n_r[:-1, 3], n_r[1:, 3])[0, 1]) < 5 / np.sqrt(M)                          # neighbouring particles


def test_many_landmarks_in_range_and_none():
    """More than 254 landmarks in range (two-byte entries in the update sweep's table, five blocks of 64 in the
    association index), a ragged particle count, and a step in which nothing is in range."""
    from slam_amd import _lib as L
    M, nl, cap = 70, 300, 340
    rng = np.random.RandomState(4)
    rad, ang = rng.uniform(1.5, 7.0, nl), rng.uniform(-np.pi, np.pi, nl)
    lm = np.stack([rad * np.cos(ang) + 0.5, rad * np.sin(ang), np.zeros(nl)], 1).astype(np.float32)
    ctx = L.Context(0, L.F64)
    ctx.fs_init(M, cap)
    orc = F.OracleFastSLAM(M, lm)
    x_gt = [0, 0, 0]
    seen = []
    for t, ut in enumerate([[8.0, 0.3], [8.0, -0.4], [4000.0, 0.001], [8.0, 0.3]]):   # the third step leaves the field
        x_gt = F.true_pose_step(x_gt, ut)
        tr, tp, tv = _truth(x_gt, lm)
        nv = int(tv.sum())
        seen.append(nv)
        ints = rng.randint(-1000, 1000, size=(M, 3, 12)).astype(np.int16)
        normals = rng.randn(M, 2 * nv)
        ru = float(rng.rand())
        idx = ctx.fs_step(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, ints, normals, ru / M, want_idx=True)
        orc.step(ut, ints, normals, ru)
        assert np.array_equal(idx, orc.last_idx), t
        _compare(ctx.fs_get_state(), orc.state(cap), t)
    assert max(seen) >= 255 and min(seen) == 0, seen


def test_one_pose_motion_sampler_and_copies():
    """FastSLAM.sample_motion_model_velocity (fast_slam1.py:167-189) for one pose under a seeded stream, against the
    oracle's pose sampler fed the same integers; Particle.copy / Feature.copy are deep."""
    from slam_amd import fast_slam1 as S
    slam = S.FastSLAM(M=4)
    np.random.seed(5)
    ints = np.stack([np.random.randint(-1000, 1000, size=12) for _ in range(3)])
    np.random.seed(5)
    x0 = np.array([[1.5, -2.0, 0.7]]).T
    out = slam.sample_motion_model_velocity(x0, [25, 0.3], dt=S.dt)
    want = F.sample_pose(x0[:, 0], [25, 0.3], ints, dt=S.dt)
    assert np.shape(out[0]) == (1,) and rel_err(np.ravel(out), want[:, 0]) < 1e-12
    flat = slam.sample_motion_model_velocity([1.5, -2.0, 0.7], [25, 0.3])
    assert np.ndim(flat[0]) == 0
    slam.control([25, 0.001])
    y = slam.Y[0]
    z = y.copy()
    assert len(z.feature) == len(y.feature) and z.x is not y.x
    if y.feature:
        z.feature[0].m[0, 0] += 1.0
        assert z.feature[0].m[0, 0] != y.feature[0].m[0, 0]


def _read_particles(ctx, which, cap):
    """Pose, weight and feature list of a few particles of a large set, through the row-packing calls of the
    multi-GPU exchange (one row per particle) instead of a copy of the whole state."""
    which = np.asarray(which, np.int64)
    ctx.fs_set_row_slots(cap)
    words = ctx.fs_row_words()
    rows = np.zeros((len(which), words), np.int64)
    prow = np.zeros((len(which), 5))
    ctx.shard_pack(which, which + 1, prow.ctypes.data)
    ctx.fs_shard_pack(which, which + 1, rows.ctypes.data)
    out = []
    for k in range(len(which)):
        y = F.Part()
        y.x = prow[k, :3].reshape(3, 1).copy()
        y.w = float(prow[k, 3])
        n = int(rows[k, 0])
        y.new_feature_id = int(rows[k, 1])
        for j in range(n):
            s = rows[k, 2 + 9 * j: 2 + 9 * (j + 1)]
            d = s[:7].view(np.float64)
            f = F.Feat(np.array([[d[0]], [d[1]]]), np.array([[d[2], d[3]], [d[4], d[5]]]))
            f.w, f.i, f.id = float(d[6]), int(s[7]), int(s[8])
            y.feature.append(f)
        out.append(y)
    return out


def test_large_configuration_subset_against_the_oracle():
    """2^20 particles x 256 landmarks (about 64 sensed per step, feature lists of 80+), the configuration that is
    timed: three steps with device-drawn numbers (ancestor indices checked against the sequential walk on the device's
    own weights), then one per-particle update with pre-drawn numbers for every particle, of which 64 particles are
    compared with the oracle run from the state the device itself held before that update."""
    from slam_amd import _lib as L
    M, nl, cap = 1 << 20, 256, 320
    rng = np.random.RandomState(9)
    lm = np.stack([rng.uniform(-8, 33, nl), rng.uniform(-24, 8, nl), np.zeros(nl)], 1).astype(np.float32)
    ctx = L.Context(0, L.F64)
    ctx.fs_init(M, cap)
    ctx.shard_set(0, M)
    x_gt = [0, 0, 0]
    for t in range(3):
        ut = [25, 0.001]
        x_gt = F.true_pose_step(x_gt, ut)
        tr, tp, tv = _truth(x_gt, lm)
        idx = ctx.fs_step(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, None, None, (0.2 + 0.3 * t) / M, seed=40 + t, want_idx=True)
        assert idx.min() >= 0 and idx.max() <= M - 1 and np.all(np.diff(idx) >= 0)
    # the state every particle-update below starts from, as the device holds it
    pick = np.sort(rng.choice(M, 64, replace=False))
    before = _read_particles(ctx, pick, cap)
    assert min(len(y.feature) for y in before) > 40
    ut = [25, -np.pi / 2 * 5]
    x_gt = F.true_pose_step(x_gt, ut)
    tr, tp, tv = _truth(x_gt, lm)
    truth = F.true_measurements(x_gt, lm)
    nv = int(tv.sum())
    assert nv > 40
    ints = rng.randint(-1000, 1000, size=(M, 3, 12)).astype(np.int16)
    normals = rng.randn(M, 2 * nv)
    ctx.fs_shard_kernel(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, ints, normals)      # fast_slam1.py:246-404 for every particle
    after = _read_particles(ctx, pick, cap)
    for k, m in enumerate(pick):
        y = before[k]
        F.particle_step(y, ut, ints[m], normals[m], truth)
        g = after[k]
        assert len(g.feature) == len(y.feature) and g.new_feature_id == y.new_feature_id, m
        assert rel_err(g.x[:, 0], y.x[:, 0]) < 1e-12 and rel_err([g.w], [y.w]) < 1e-11, m
        for fg, fo in zip(g.feature, y.feature):
            assert (fg.i, fg.id) == (fo.i, fo.id), m
            assert rel_err(fg.m, fo.m) < 1e-12 and rel_err([fg.w], [fo.w]) < 1e-11, m
            assert np.all(np.abs(fg.Cov - fo.Cov) <= 1e-11 * np.max(np.abs(fo.Cov))), m
    ctx.close()
