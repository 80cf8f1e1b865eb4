#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the reference.

Runs ONLY in the build container (needs /root/reference, read-only).  The
reference's Python modules are imported as they lie (a stub `cv2` module is
registered because `mc_localization.py:4` imports it at module top but only
uses it under `__main__`), driven with a seeded legacy `np.random` stream, and
the inputs/outputs of every stage are written as compressed .npz files.  No
reference source text is stored: fixtures are numbers only.

Usage:  MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
import contextlib
import io
import os
import runpy
import sys
import types
import warnings

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

REF = "/root/reference/08_grid_and_montecarlo_localization"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
OUT = os.path.abspath(OUT)

sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, REF)
import mc_localization as ref_mcl  # noqa: E402  (the reference module itself)
import range_sensor as ref_rs  # noqa: E402
import robot as ref_robot  # noqa: E402


def load_ref_map():
    """G channel of grid_map.png == cv2.imread(...)[:, :, 1] (mc_localization.py:304-307)."""
    im = Image.open(os.path.join(REF, "grid_map.png")).convert("RGB")
    return np.ascontiguousarray(np.array(im)[:, :, 1])


def synth_map(n, seed, n_seg):
    """Synthetic {0,255} map: 2-cell border wall + random axis-aligned segments."""
    rng = np.random.RandomState(seed)
    m = np.full((n, n), 255, np.uint8)
    m[:2, :] = 0
    m[-2:, :] = 0
    m[:, :2] = 0
    m[:, -2:] = 0
    for _ in range(n_seg):
        length = rng.randint(8, 65)
        thick = rng.randint(1, 3)
        i = rng.randint(0, n)
        j = rng.randint(0, n)
        if rng.randint(0, 2):
            m[i:i + thick, j:j + length] = 0
        else:
            m[i:i + length, j:j + thick] = 0
    # a few grey cells: neither obstacle (==0) nor free (==255)
    for _ in range(n // 4):
        m[rng.randint(2, n - 2), rng.randint(2, n - 2)] = 195
    return m


def make_ref(map_g, M, sensor_angles, x_gt):
    """Reference filter object with overridden size / beam set (attributes only)."""
    obj = ref_mcl.MCLocalization(np.array(x_gt), map_g)
    obj.M = M
    obj.X = np.zeros([M, 3])
    obj.w = np.ones(M) / M
    if sensor_angles is not None:
        obj.sensor_angles = list(sensor_angles)
        obj.sensor_ranges = [0] * len(sensor_angles)
        obj.true_ranges = [0] * len(sensor_angles)
    ref_mcl.mcl = obj  # sample() reads the module-global `mcl` (mc_localization.py:223)
    return obj


def harness_hits(obj, pose):
    """Row-major argmin obstacle cell per beam, following measure()'s predicate
    with the reference's own obstacle LUT (harness restatement; the reference
    does not return hit cells)."""
    x, y, th = pose[0], pose[1], pose[2]
    H, W = obj.map.shape
    obs = obj.obstacles
    B = len(obj.sensor_angles)
    hits = np.full(B, -1, np.int64)
    if len(obs) == 0:
        return hits
    dx = obs[:, 0] - x
    dy = obs[:, 1] - y
    r2 = dx ** 2 + dy ** 2
    keep = np.where(r2 <= ref_mcl.max_sensor_range ** 2)[0]
    dxv, dyv = dx[keep], dy[keep]
    ang = np.arctan2(dyv, dxv)
    ang[ang < 0.0] += np.pi * 2.0
    thr = 5 * np.pi / 180.0
    cols = np.rint(obs[:, 0] / np.float32(ref_mcl.res_x)).astype(np.int64)
    rows = np.rint(obs[:, 1] / np.float32(ref_mcl.res_y)).astype(np.int64)
    for k in range(B):
        beam = obj.sensor_angles[k] * np.pi / 180.0 + th
        d = abs(beam - ang)
        sel = np.where((d <= thr) | ((d >= np.pi * 2 - thr) & (d <= np.pi * 2 + thr)))[0]
        if len(sel):
            rr = dxv[sel] ** 2 + dyv[sel] ** 2
            g = keep[sel[int(np.argmin(rr))]]
            hits[k] = rows[g] * W + cols[g]
    return hits


def ref_indices(obj, w_before, state_before):
    """Resample indices, produced by the reference's own loop on an index-tagged X."""
    keepX, keepw = obj.X, obj.w
    st_after = np.random.get_state()
    np.random.set_state(state_before)
    obj.X = np.stack([np.arange(obj.M, dtype=np.float64), np.zeros(obj.M), np.zeros(obj.M)], 1)
    obj.w = w_before.copy()
    obj.resample()
    idx = obj.X[:, 0].astype(np.int64)
    obj.X, obj.w = keepX, keepw
    np.random.set_state(st_after)
    return idx


def run_steps(obj, controls, dt, want_detail=True):
    """Drive sample()+resample() and capture every stage."""
    rec = []
    M = obj.M
    B = len(obj.sensor_angles)
    for (v, w) in controls:
        s = {}
        s["ut"] = np.array([v, w], np.float64)
        s["dt"] = np.float64(dt)
        s["X_in"] = obj.X.copy()
        s["x_gt_in"] = np.array(obj.x_gt).copy()
        s["x_gt_in_is_f32"] = np.array(np.asarray(obj.x_gt).dtype == np.float32)
        st = np.random.get_state()
        obj.sample([v, w], dt)
        st_after = np.random.get_state()
        # the raw integers sample() consumed: same stream, drawn batched
        np.random.set_state(st)
        s["noise"] = np.random.random_integers(-1000, 1000, size=(M, 3, 12)).astype(np.int16)
        assert all(np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b
                   for a, b in zip(np.random.get_state(), st_after)), "noise stream mismatch"
        np.random.set_state(st_after)
        s["X_motion"] = obj.X.copy()
        s["x_gt_out"] = np.array(obj.x_gt, np.float64).copy()
        s["z_meas"] = np.array(obj.sensor_ranges, np.float64)
        s["w_raw"] = obj.w.copy()
        if want_detail:
            rng_ = np.zeros((M, B))
            pk = np.zeros((M, B))
            hits = np.zeros((M, B), np.int64)
            pm = np.zeros(M)
            for m in range(M):
                rr = obj.measure(obj.X[m])
                rng_[m] = rr
                hits[m] = harness_hits(obj, obj.X[m])
                for k in range(B):
                    pk[m, k] = obj.range_sensor.compute_px(rr[k], obj.sensor_ranges[k])
                u = int(obj.X[m, 0] / ref_mcl.res_x + 0.5)
                vv = int(obj.X[m, 1] / ref_mcl.res_y + 0.5)
                ok = 0 <= u < obj.map.shape[1] and 0 <= vv < obj.map.shape[0] and obj.map[vv, u] == 255
                pm[m] = 1.0 if ok else 0.0
            s["ranges"] = rng_
            s["hits"] = hits
            s["p_beam"] = pk
            s["pm"] = pm
            s["hits_gt"] = harness_hits(obj, obj.x_gt)
        s["w_sum"] = np.float64(np.sum(obj.w))
        st = np.random.get_state()
        w_before = obj.w.copy()
        try:
            obj.resample()
            ok = True
        except IndexError:
            ok = False
        s["resample_ok"] = np.array(ok)
        if ok:
            np_state_after = np.random.get_state()
            np.random.set_state(st)
            s["r"] = np.float64(np.random.rand() * 1 / M)
            np.random.set_state(np_state_after)
            s["idx"] = ref_indices(obj, w_before, st)
            s["X_out"] = obj.X.copy()
            s["w_out"] = obj.w.copy()
        rec.append(s)
        if not ok:
            break
    return rec


def save_steps(name, rec, extra):
    flat = dict(extra)
    flat["n_steps"] = np.array(len(rec))
    for t, s in enumerate(rec):
        for k, v in s.items():
            flat["s%d_%s" % (t, k)] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **flat)
    print("wrote", name, "steps", len(rec))


def case_reference_demo():
    """The reference demo: grid_map.png, M=500, 13 beams (mc_localization.py:303-331)."""
    map_g = load_ref_map()
    np.save(os.path.join(OUT, "grid_map_g.npy"), map_g)
    for seed in (0, 1, 2):
        np.random.seed(seed)
        x0 = np.array([20 * ref_mcl.res_x, 30 * ref_mcl.res_y, 0], dtype=np.float32)
        obj = make_ref(map_g, 500, None, x0)
        obj.x_gt = x0
        obj.initialize(x0)
        nsteps = 6 if seed == 0 else 2
        ctrl = [(1.5, 0)] * nsteps
        if seed == 2:
            ctrl = [(1.5, 0.5), (1.0, -0.7)]
        rec = run_steps(obj, ctrl, 0.1)
        save_steps("demo_seed%d" % seed, rec,
                   {"sensor_angles": np.array(obj.sensor_angles, np.float64),
                    "x0": x0, "seed": np.array(seed)})


def case_synth(name, n, seed, M, B, nseg, controls):
    map_g = synth_map(n, seed, nseg)
    np.random.seed(seed)
    rng = np.random.RandomState(seed + 1000)
    angles = [k * 360.0 / B for k in range(B)]
    free = np.argwhere(map_g == 255)
    pick = free[rng.randint(0, len(free), M)]
    X = np.zeros((M, 3))
    X[:, 0] = pick[:, 1] * 0.15 + rng.uniform(-0.075, 0.075, M)
    X[:, 1] = pick[:, 0] * 0.15 + rng.uniform(-0.075, 0.075, M)
    X[:, 2] = rng.uniform(0, 2 * np.pi, M)
    # spread headings beyond one turn for a few particles: the beam angle is
    # never wrapped by the reference (mc_localization.py:143-149)
    X[: M // 16, 2] += rng.choice([-4 * np.pi, -2 * np.pi, 2 * np.pi, 4 * np.pi], M // 16)
    # half of the cloud sits near the true pose so that weights are not all ~0
    X[M // 2:, 0] = X[0, 0] + rng.normal(0, 0.05, M - M // 2)
    X[M // 2:, 1] = X[0, 1] + rng.normal(0, 0.05, M - M // 2)
    X[M // 2:, 2] = X[0, 2] + rng.normal(0, 0.02, M - M // 2)
    obj = make_ref(map_g, M, angles, X[0].copy())
    obj.X = X.copy()
    obj.x_gt = X[0].copy()
    rec = run_steps(obj, controls, 0.1)
    save_steps(name, rec, {"map": map_g, "sensor_angles": np.array(angles), "seed": np.array(seed)})


def case_test_robot():
    """Run the reference's own test script (test_robot.py) with Robot.measure
    instrumented; capture pose, map and the 13 ranges at every call plus the
    OK/error lines it prints."""
    calls = []
    orig = ref_robot.Robot.measure

    def spy(self, xt, map, noise=True):
        out = orig(self, xt, map, noise)
        calls.append((np.array(xt, np.float64).ravel().copy(), np.array(map).copy(),
                      np.array(out, np.float64).copy()))
        return out

    ref_robot.Robot.measure = spy
    buf = io.StringIO()
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        with contextlib.redirect_stdout(buf):
            runpy.run_path(os.path.join(REF, "test_robot.py"), run_name="__main__")
    finally:
        os.chdir(cwd)
        ref_robot.Robot.measure = orig
    lines = buf.getvalue().splitlines()
    # replay: attribute each "Beam k: OK" line to the measure() call that preceded it
    checks = []  # (call index, beam id, ok)
    # the script prints in program order; re-run the pairing by counting calls:
    # we record the call count at each print through a second instrumented pass
    calls2 = []
    marks = []

    def spy2(self, xt, map, noise=True):
        out = orig(self, xt, map, noise)
        calls2.append(1)
        return out

    class Tap(io.StringIO):
        def write(self, s):
            if s.startswith("Beam"):
                marks.append((len(calls2) - 1, s.strip()))
            return super().write(s)

    ref_robot.Robot.measure = spy2
    os.chdir(REF)
    try:
        with contextlib.redirect_stdout(Tap()):
            runpy.run_path(os.path.join(REF, "test_robot.py"), run_name="__main__")
    finally:
        os.chdir(cwd)
        ref_robot.Robot.measure = orig
    for ci, text in marks:
        beam = int(text.split()[1].rstrip(":"))
        checks.append((ci, beam, text.endswith("OK")))
    poses = np.stack([c[0] for c in calls])
    maps = np.stack([c[1] for c in calls])
    ranges = np.stack([c[2] for c in calls])
    np.savez_compressed(os.path.join(OUT, "test_robot.npz"), poses=poses, maps=maps, ranges_f32path=ranges,
                        checks=np.array(checks, np.int64),
                        sensor_angles=np.array([0, 15, 30, 45, 60, 75, 90, 345, 330, 315, 300, 285, 270], np.float64))
    print("wrote test_robot: calls", len(calls), "checks", len(checks), "all ok", all(c[2] for c in checks),
          "printed", sum(1 for l in lines if l.startswith("Beam")))


def case_compute_px():
    rs = ref_rs.RangeSensor()
    rs.z_max = 5
    zs = [0.0, 1e-9, 0.05, 0.15, 0.5, 1.0, 1.7320508, 2.5, 3.3, 4.0, 4.85, 4.999, 4.9999985, 4.9999995, 5.0]
    zt = [1e-3, 0.1, 0.5, 1.0, 2.5, 3.3, 4.0, 4.999, 5.0]
    rng = np.random.RandomState(7)
    zs = zs + list(rng.uniform(0, 5, 200))
    zt = zt + list(rng.uniform(0.01, 5, 40))
    Z, ZT = np.meshgrid(np.array(zs), np.array(zt), indexing="ij")
    P = np.zeros_like(Z)
    for i in range(Z.shape[0]):
        for j in range(Z.shape[1]):
            P[i, j] = rs.compute_px(np.float64(Z[i, j]), np.float64(ZT[i, j]))
    np.savez_compressed(os.path.join(OUT, "compute_px.npz"), z=Z, z_true=ZT, p=P,
                        w4=np.array([rs.w_hit, rs.w_short, rs.w_max, rs.w_rand]),
                        sigma_hit=np.float64(rs.sigma_hit), lambda_short=np.float64(rs.lambda_short),
                        z_max=np.float64(rs.z_max))
    print("wrote compute_px", P.shape)


def case_resample_only():
    """Weights -> indices through the reference loop, M = 4096 and shapes that
    stress the CDF (few dominant weights, zeros, equal weights)."""
    out = {}
    n = 0
    for seed, M, kind in [(11, 4096, "rand"), (12, 4096, "peaky"), (13, 4096, "zeros"), (14, 512, "equal"),
                          (15, 4096, "tiny"), (16, 1000, "rand")]:
        rng = np.random.RandomState(seed)
        if kind == "rand":
            w = rng.rand(M)
        elif kind == "peaky":
            w = rng.rand(M) ** 40
        elif kind == "zeros":
            w = rng.rand(M) * (rng.rand(M) < 0.3)
        elif kind == "equal":
            w = np.ones(M)
        else:
            w = np.exp(-rng.rand(M) * 600)
        map_g = np.full((4, 4), 255, np.uint8)
        obj = make_ref(map_g, M, None, np.zeros(3))
        obj.X = np.stack([np.arange(M, dtype=np.float64), np.zeros(M), np.zeros(M)], 1)
        obj.w = w.copy()
        np.random.seed(seed)
        st = np.random.get_state()
        try:
            obj.resample()
        except IndexError:
            continue
        np.random.set_state(st)
        r = np.random.rand() * 1 / M
        out["c%d_w" % n] = w
        out["c%d_r" % n] = np.float64(r)
        out["c%d_idx" % n] = obj.X[:, 0].astype(np.int64)
        out["c%d_w_out" % n] = obj.w.copy()
        n += 1
    out["n_cases"] = np.array(n)
    np.savez_compressed(os.path.join(OUT, "resample_only.npz"), **out)
    print("wrote resample_only", n)


def case_motion_only():
    """sample_motion_model_velocity over many poses/controls incl. w_hat == 0 and negative coordinates."""
    map_g = load_ref_map()
    obj = make_ref(map_g, 8, None, np.zeros(3))
    rng = np.random.RandomState(5)
    N = 2000
    X = np.stack([rng.uniform(-1, 16, N), rng.uniform(-1, 16, N), rng.uniform(-7, 7, N)], 1)
    ctrls = [(1.5, 0.0), (1.5, 0.5), (0.0, 0.0), (-0.7, -1.2), (0.0, 0.9)]
    np.random.seed(99)
    st = np.random.get_state()
    Xo = np.zeros((len(ctrls), N, 3))
    pm = np.zeros((len(ctrls), N))
    for c, ut in enumerate(ctrls):
        for m in range(N):
            xt, p = obj.sample_motion_model_velocity(X[m], list(ut), noise=True, dt=0.1)
            Xo[c, m] = xt
            pm[c, m] = p
    np.random.set_state(st)
    noise = np.random.random_integers(-1000, 1000, size=(len(ctrls), N, 3, 12)).astype(np.int16)
    # force exact zero angular-velocity draws: all-zero integers with w = 0
    Z = np.zeros((N, 3))
    Zp = np.zeros(N)

    class _Zero:
        pass
    saved = ref_mcl.sample
    ref_mcl.sample = lambda b: 0.0 * b
    try:
        for m in range(N):
            xt, p = obj.sample_motion_model_velocity(X[m], [1.5, 0.0], noise=True, dt=0.1)
            Z[m] = xt
            Zp[m] = p
    finally:
        ref_mcl.sample = saved
    np.savez_compressed(os.path.join(OUT, "motion_only.npz"), X=X, controls=np.array(ctrls), noise=noise,
                        X_out=Xo, pm=pm, X_out_zero_noise=Z, pm_zero_noise=Zp, alpha=np.array(ref_mcl.alpha, np.float64))
    print("wrote motion_only")


def case_amcl():
    """Augmented MCL (mc_localization2.py): sample with noisy predicted ranges + running weight
    means, resample with random-pose injection.  The random draws are re-derived from the saved
    generator state in the order the reference consumes them and checked against its end state."""
    import mc_localization2 as ref2
    map_g = load_ref_map()
    for seed, M, nsteps in ((0, 300, 8), (4, 200, 4)):
        np.random.seed(seed)
        x0 = np.array([20 * ref2.res_x, 30 * ref2.res_y, 0], dtype=np.float32)
        obj = ref2.MCLocalization(np.array(x0), map_g)
        obj.M = M
        obj.X = np.zeros([M, 3])
        obj.w = np.ones(M) / M
        ref2.mcl = obj
        obj.x_gt = x0
        obj.initialize(x0)
        B = len(obj.sensor_angles)
        rec = []
        for t in range(nsteps):
            s = {}
            s["X_in"] = obj.X.copy()
            s["x_gt_in"] = np.array(obj.x_gt).copy()
            s["x_gt_in_is_f32"] = np.array(np.asarray(obj.x_gt).dtype == np.float32)
            s["w_slow_in"] = np.float64(obj.w_slow)
            s["w_fast_in"] = np.float64(obj.w_fast)
            st = np.random.get_state()
            with contextlib.redirect_stdout(io.StringIO()):
                obj.sample([1.5, 0], 0.1)
            st_after = np.random.get_state()
            np.random.set_state(st)
            ints = np.zeros((M, 3, 12), np.int16)
            rn = np.zeros((M, B))
            for m in range(M):
                for j in range(3):
                    ints[m, j] = np.random.randint(-1000, 1000, size=12)
                for k in range(B):
                    rn[m, k] = np.random.randn() * 0.15
            assert all(np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b
                       for a, b in zip(np.random.get_state(), st_after)), "sample stream mismatch"
            s["noise"], s["range_noise"] = ints, rn
            s["X_motion"] = obj.X.copy()
            s["x_gt_out"] = np.array(obj.x_gt, np.float64)
            s["z_meas"] = np.array(obj.sensor_ranges, np.float64)
            s["w_raw"] = obj.w.copy()
            s["w_avg"] = np.float64(obj.w_avg)
            s["w_slow"] = np.float64(obj.w_slow)
            s["w_fast"] = np.float64(obj.w_fast)
            s["w_sum"] = np.float64(np.sum(obj.w))
            if t >= nsteps - 3:
                # state input of resample(): a slow mean above the fast one makes it inject
                obj.w_slow = obj.w_fast / (0.7 if t == nsteps - 3 else 0.2)
            s["w_slow_rs"] = np.float64(obj.w_slow)
            s["w_fast_rs"] = np.float64(obj.w_fast)
            st = np.random.get_state()
            with contextlib.redirect_stdout(io.StringIO()):
                obj.resample()
            st_after = np.random.get_state()
            np.random.set_state(st)
            s["r"] = np.float64(np.random.rand() * 1 / M)
            with np.errstate(all="ignore"):
                p_random = max(0.0, 1 - obj.w_fast / obj.w_slow)
            rows, poses = [], []
            for m in range(M):
                if np.random.rand() <= p_random:
                    while True:
                        x = np.random.rand() * ref2.max_x
                        y = np.random.rand() * ref2.max_y
                        u, v = int(x / ref2.res_x + 0.5), int(y / ref2.res_y + 0.5)
                        if not (u >= map_g.shape[1] or v >= map_g.shape[0] or map_g[v, u] < 255):
                            break
                    rows.append(m)
                    poses.append([x, y, np.random.rand() * np.pi * 2.0])
            assert all(np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b
                       for a, b in zip(np.random.get_state(), st_after)), "resample stream mismatch"
            s["p_random"] = np.float64(p_random)
            s["inj_rows"] = np.array(rows, np.int64)
            s["inj_poses"] = np.array(poses, np.float64).reshape(-1, 3)
            s["X_out"] = obj.X.copy()
            s["w_out"] = obj.w.copy()
            rec.append(s)
        print("amcl seed", seed, "injections per step", [len(x["inj_rows"]) for x in rec])
        save_steps("amcl_seed%d" % seed, rec, {"sensor_angles": np.array(obj.sensor_angles, np.float64),
                                               "x0": x0, "seed": np.array(seed)})


def case_grid():
    """Grid localisation (robot.py / grid_localization.py): the reference's Robot on its own
    grid_map0.png, predict() + update() under a seeded np.random, every stage captured."""
    im = Image.open(os.path.join(REF, "grid_map0.png")).convert("RGB")
    grid = np.ascontiguousarray(np.array(im)[:, :, 2])      # cv2.imread(...)[:, :, 0] is the blue channel
    for name, seed, controls in (("grid_seed0", 0, [(1.5, 0), (1.5, 0)]), ("grid_seed1", 1, [(1.5, 0.5)])):
        np.random.seed(seed)
        x_pose = np.array([[20 * ref_robot.res_x, 30 * ref_robot.res_y, 0]], dtype=np.float32).T
        with contextlib.redirect_stdout(io.StringIO()):
            robot = ref_robot.Robot(grid, x_pose)
        out = {"map": grid, "sensor_angles": np.array(robot.sensor_angles, np.float64), "seed": np.array(seed),
               "pk0": robot.pk.copy(), "n_steps": np.array(len(controls))}
        for t, ut in enumerate(controls):
            draws = []
            first = []
            orig_random = np.random.random
            orig_pm = ref_robot.Robot.predict_measure

            def spy_random(*a, **k):
                v = orig_random(*a, **k)
                draws.append(v)
                return v

            def spy_pm(self, xt, map):
                r = orig_pm(self, xt, map)
                if not first:
                    first.append([np.float64(v) for v in r])
                return r

            pre = {"pk_in": robot.pk.copy(), "x_gt_in": robot.x_gt.copy()}
            with contextlib.redirect_stdout(io.StringIO()):
                robot.predict(list(ut))
                np.random.random = spy_random
                ref_robot.Robot.predict_measure = spy_pm
                try:
                    pk_pred = robot.pk_pred.copy()
                    robot.update()
                finally:
                    np.random.random = orig_random
                    ref_robot.Robot.predict_measure = orig_pm
            rec = dict(pre)
            rec.update({"ut": np.array(ut, np.float64), "x_gt_out": robot.x_gt.copy(), "pk_pred": pk_pred,
                        "uniforms": np.array(draws), "z_true": np.array(first[0]),
                        "z_meas": np.array([float(v) for v in robot.sensor_ranges]), "pk_out": robot.pk.copy()})
            for k, v in rec.items():
                out["s%d_%s" % (t, k)] = v
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print("wrote", name, "steps", len(controls), "pk max", float(robot.pk.max()), "dtype", robot.pk.dtype)


def case_fastslam():
    """FastSLAM 1.0 (13_fast_slam/fast_slam1.py): the reference's FastSLAM object driven by its own
    __main__ controls under a seeded np.random, plotting switched off; the random numbers every
    step consumed and the full particle state after it are captured."""
    import importlib.util
    fs_dir = "/root/reference/13_fast_slam"
    spec = importlib.util.spec_from_file_location("ref_fast_slam1", os.path.join(fs_dir, "fast_slam1.py"))
    fs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fs)
    fs.FastSLAM.plot = lambda self: None
    tm = fs.tm

    def control_at(t):     # the schedule of fast_slam1.py:539-559
        turn = [25, -np.pi / 2 * 5]
        straight = [25, 0.001]
        if t <= 7 * tm: return straight
        if t <= 9 * tm: return turn
        if t <= 10 * tm: return straight
        if t <= 12 * tm: return turn
        if t <= 17 * tm: return straight
        if t <= 19 * tm: return turn
        if t <= 20 * tm: return straight
        if t <= 22 * tm: return turn
        return straight

    for name, seed, n_steps in (("fastslam_seed0", 0, 26), ("fastslam_seed3", 3, 12)):
        np.random.seed(seed)
        slam = fs.FastSLAM()
        M = slam.M
        out = {"seed": np.array(seed), "n_steps": np.array(n_steps), "M": np.array(M),
               "landmarks": np.array(fs.landmarks, np.float64), "dt": np.array(fs.dt)}
        cap = 0
        recs = []
        for t in range(n_steps):
            ints, normals, uniforms = [], [], []
            o_randint, o_randn, o_rand = np.random.randint, np.random.randn, np.random.rand

            def spy_randint(*a, **k):
                v = o_randint(*a, **k); ints.append(np.array(v)); return v

            def spy_randn(*a, **k):
                v = o_randn(*a, **k); normals.append(float(v)); return v

            def spy_rand(*a, **k):
                v = o_rand(*a, **k); uniforms.append(float(v)); return v

            np.random.randint, np.random.randn, np.random.rand = spy_randint, spy_randn, spy_rand
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    ut = control_at(t)
                    slam.control(ut)
            finally:
                np.random.randint, np.random.randn, np.random.rand = o_randint, o_randn, o_rand
            rec = {"ut": np.array(ut, np.float64), "ints": np.array(ints).reshape(M, 3, 12).astype(np.int16),
                   "normals": np.array(normals).reshape(M, -1), "r_uniform": np.array(uniforms),
                   "x_gt": np.array(slam.x_gt[-1], np.float64),
                   "pose": np.stack([y.x[-1][:, 0] for y in slam.Y]),
                   "w": np.array([y.w for y in slam.Y]), "nfeat": np.array([len(y.feature) for y in slam.Y]),
                   "new_id": np.array([y.new_feature_id for y in slam.Y])}
            cap = max(cap, int(rec["nfeat"].max()))
            rec["_feat"] = [[(f.m[:, 0].copy(), f.Cov.copy(), f.i, f.w, f.id, bool(f.updated)) for f in y.feature] for y in slam.Y]
            recs.append(rec)
        for t, rec in enumerate(recs):
            feats = rec.pop("_feat")
            fm = np.zeros((M, cap, 2)); fc = np.zeros((M, cap, 2, 2)); fi = np.zeros((M, cap), np.int64)
            fw = np.zeros((M, cap)); fid = np.full((M, cap), -1, np.int64); fu = np.zeros((M, cap), bool)
            for k, fl in enumerate(feats):
                for j, (m, C, i, w, idn, upd) in enumerate(fl):
                    fm[k, j], fc[k, j], fi[k, j], fw[k, j], fid[k, j], fu[k, j] = m, C, i, w, idn, upd
            rec.update({"f_m": fm, "f_cov": fc, "f_i": fi, "f_w": fw, "f_id": fid, "f_updated": fu})
            for k, v in rec.items():
                out["s%d_%s" % (t, k)] = v
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print("wrote", name, "steps", n_steps, "max features", cap, "normals per particle",
              [r["normals"].shape[1] for r in recs][:8])


def _load_script(path, name, shadow=()):
    """Runs one of the reference's stand-alone scripts as a module (they have no __main__ guard: the
    demo they contain runs once, headless, output discarded) and returns it.  `shadow`: names of
    already-imported modules a same-named sibling of the script must replace while it loads."""
    import importlib.util
    saved = {k: sys.modules.pop(k) for k in shadow if k in sys.modules}
    sys.path.insert(0, os.path.dirname(path))
    try:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        with contextlib.redirect_stdout(io.StringIO()):
            spec.loader.exec_module(mod)
        import matplotlib.pyplot as plt
        plt.close("all")
    finally:
        sys.path.remove(os.path.dirname(path))
        for k in shadow:
            sys.modules.pop(k, None)
        sys.modules.update(saved)
    return mod


def case_particle_filter04():
    """The low-variance sampler of the chapter-4 script (04_nonparametric_filters/particle_filter.py:58-81):
    its ParticleFilter.resample() on weight vectors of several sizes, then three iterations of the
    script's own loop body (sample_gaussian, importance weights, resample) for the drop-in class."""
    np.random.seed(404)
    pf = _load_script("/root/reference/04_nonparametric_filters/particle_filter.py", "ref_particle_filter04")
    out = {}
    n = 0
    for seed, M, kind in [(41, 50, "rand"), (42, 50, "peaky"), (43, 777, "zeros"), (44, 2048, "rand"), (45, 64, "equal"),
                          (46, 1, "rand")]:
        rng = np.random.RandomState(seed)
        w = {"rand": rng.rand(M), "peaky": rng.rand(M) ** 30, "zeros": rng.rand(M) * (rng.rand(M) < 0.4),
             "equal": np.ones(M)}[kind]
        pf.M = M                                   # the method reads the module's M (:71, :87)
        obj = pf.ParticleFilter(M, np.ones(pf.x_max) / pf.x_max)
        X = rng.uniform(0, pf.x_max - 1, M)
        obj.X = list(X)
        obj.w = w.copy()
        np.random.seed(seed)
        st = np.random.get_state()
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                obj.resample()
        except IndexError:
            continue
        np.random.set_state(st)
        r = np.random.rand() * 1 / M
        out["c%d_w" % n] = w
        out["c%d_X" % n] = X
        out["c%d_r" % n] = np.float64(r)
        out["c%d_X_out" % n] = np.array(obj.X, np.float64)
        out["c%d_w_out" % n] = np.array(obj.w, np.float64)
        out["c%d_px" % n] = np.array(obj.px, np.float64)
        n += 1
    out["n_cases"] = np.array(n)
    out["x_max"] = np.array(pf.x_max)
    # the script's loop (:110-162) without the plotting, three iterations
    M = 50
    pf.M = M
    np.random.seed(7)
    flt = pf.ParticleFilter(M, np.ones(pf.x_max) / pf.x_max)
    for it in range(3):
        with contextlib.redirect_stdout(io.StringIO()):
            if it == 0:
                flt.sample_uniform()
            else:
                flt.sample_gaussian()
            pz = np.zeros([pf.x_max])
            for m in range(M):
                z = int(np.clip(flt.X[m], a_min=0, a_max=pf.x_max - 1) + 0.5)
                pz[z] += 1
            pz /= M
            for m in range(M):
                z = int(np.clip(flt.X[m], a_min=0, a_max=pf.x_max - 1) + 0.5)
                flt.w[m] = pf.normal_dist(pf.x_obj, 1, z) * pz[z] / (flt.px[z] + 1e-6)
            out["loop%d_X_in" % it] = np.array(flt.X, np.float64)
            out["loop%d_w_in" % it] = np.array(flt.w, np.float64)
            flt.resample()
        out["loop%d_X" % it] = np.array(flt.X, np.float64)
        out["loop%d_w" % it] = np.array(flt.w, np.float64)
        out["loop%d_px" % it] = np.array(flt.px, np.float64)
    out["x_obj"] = np.array(pf.x_obj)
    np.savez_compressed(os.path.join(OUT, "particle_filter04.npz"), **out)
    print("wrote particle_filter04", n)


def case_motion05():
    """Table 5.3 as the chapter-5 script states it (05_robot_motion/sample_motion_model_velocity.py:90-108;
    no straight-line branch, alpha of :11): its sample_motion_model_velocity over many poses and
    controls with the integers it drew, and with the sampler forced to zero at w = 0 (division by zero)."""
    np.random.seed(505)
    mv = _load_script("/root/reference/05_robot_motion/sample_motion_model_velocity.py", "ref_motion05",
                      shadow=("robot",))
    rng = np.random.RandomState(6)
    N = 1500
    X = np.stack([rng.uniform(-3, 12, N), rng.uniform(-3, 12, N), rng.uniform(-7, 7, N)], 1)
    ctrls = [(10.0, np.pi / 4 * 10), (1.5, 0.5), (1.5, 0.0), (-0.7, -1.2), (0.0, 0.9)]
    np.random.seed(98)
    st = np.random.get_state()
    Xo = np.zeros((len(ctrls), N, 3))
    for c, ut in enumerate(ctrls):
        for m in range(N):
            Xo[c, m] = mv.sample_motion_model_velocity(np.array(ut), X[m], dt=mv.delta_t)
    np.random.set_state(st)
    noise = np.random.random_integers(-1000, 1000, size=(len(ctrls), N, 3, 12)).astype(np.int16)
    Z = np.zeros((N, 3))
    saved = mv.sample
    mv.sample = lambda b: np.float64(0.0) * b
    try:
        for m in range(N):
            Z[m] = mv.sample_motion_model_velocity(np.array([1.5, 0.0]), X[m], dt=mv.delta_t)
    finally:
        mv.sample = saved
    # the Table 5.1 density on a few triples (host helper of the drop-in module)
    T = 64
    xt = np.stack([rng.uniform(0, 3, T), rng.uniform(0, 3, T), rng.uniform(-3, 3, T)], 1)
    xt1 = np.stack([rng.uniform(0, 3, T), rng.uniform(0, 3, T), rng.uniform(-3, 3, T)], 1)
    dens = np.array([mv.motion_model_velocity(xt[i], np.array([10.0, np.pi / 4 * 10]), xt1[i], dt=mv.delta_t) for i in range(T)])
    np.savez_compressed(os.path.join(OUT, "motion05.npz"), X=X, controls=np.array(ctrls), noise=noise, X_out=Xo,
                        X_out_zero=Z, alpha=np.array(mv.alpha, np.float64), dt=np.array(mv.delta_t),
                        dens_xt=xt, dens_xt1=xt1, dens=dens)
    print("wrote motion05, NaN/inf rows with w_hat == 0:", int(np.sum(~np.isfinite(Z).all(1))))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["demo", "robot", "px", "resample", "motion", "synth", "amcl", "grid", "fastslam", "pf04",
                             "motion05"]
    if "pf04" in which:
        case_particle_filter04()
    if "motion05" in which:
        case_motion05()
    if "amcl" in which:
        case_amcl()
    if "grid" in which:
        case_grid()
    if "fastslam" in which:
        case_fastslam()
    if "demo" in which:
        case_reference_demo()
    if "robot" in which:
        case_test_robot()
    if "px" in which:
        case_compute_px()
    if "resample" in which:
        case_resample_only()
    if "motion" in which:
        case_motion_only()
    if "synth" in which:
        case_synth("synth128_b13", 128, 21, 256, 13, 12, [(1.5, 0.0), (1.5, 0.5)])
        case_synth("synth256_b180", 256, 22, 768, 180, 48, [(1.5, 0.0)])
        case_synth("synth256_b360", 256, 23, 192, 360, 48, [(1.5, 0.5)])
