"""One filter over two ranks sharing the test box's single GPU (gloo carries the collectives):
the sharded update through the C ABI must equal the single-context update bit for bit."""
import os

import numpy as np
import pytest

from shard_helpers import run_ranks
from util import GOLDEN

pytestmark = pytest.mark.gpu

ALPHA = [4, 2, 4, 4, 4, 2]
ANGLES = [0, 15, 30, 45, 60, 75, 90, 345, 330, 315, 300, 285, 270]


def _setup(dtype, flat=False):
    from slam_amd import _lib
    grid = np.load(os.path.join(GOLDEN, "grid_map_g.npy"))
    if flat:
        grid = np.full_like(grid, 255)          # no obstacle: every particle on the map gets the same weight
    ctx = _lib.Context(0, dtype)
    ctx.set_map(grid, 0.15, 0.15)
    w4 = np.array([0.97, 0.005, 0.020, 0.005])
    ctx.set_sensor(ANGLES, 5, 0.15, 1.0, w4 / w4.sum(), 5 * np.pi / 180.0)
    return ctx


def _inputs(M):
    rng = np.random.RandomState(3)
    X = np.stack([rng.normal(3.0, 0.5, M), rng.normal(4.5, 0.5, M), rng.normal(0, 0.6, M)], 1)
    noise = rng.randint(-1000, 1001, size=(M, 3, 12)).astype(np.int16)
    return X, noise, rng.rand() / M


def _rank_fn(rank, world, dist, sizes, dtype, mode, z_meas, use_device_noise, device_path=None):
    from slam_amd.sharded import ShardedFilter
    if device_path:
        import torch
        torch.cuda.set_device(0)
    M = sum(sizes)
    X, noise, r = _inputs(M)
    first = sum(sizes[:rank])
    sl = slice(first, first + sizes[rank])
    ctx = _setup(dtype)
    ctx.set_particles(X[sl])
    f = ShardedFilter(ctx, dist, rank, world, sizes[rank], weight_mode=mode, device_path=device_path)
    assert f.device_path == bool(device_path)
    f.step(1.5, 0.0, 0.1, ALPHA, z_meas, r, seed=99, noise_ints=None if use_device_noise else noise[sl], want_idx=True)
    Xo, wo, _ = ctx.get_particles()
    return {"idx": f.last_idx, "X": Xo, "w": wo, "total": f.total}


@pytest.mark.parametrize("sizes,devnoise", [((16384, 8192), False), ((8192, 8192, 11000), True)])
def test_device_driven_step_between_processes_equals_one_context(sizes, devnoise):
    """The device-driven step between real processes (one per rank, sharing the box's GPU; gloo carries the
    collectives on host copies of the device tensors): the calls torch.distributed gets -- gathered tensors, split
    sizes of the all-to-all -- are the ones an RCCL run makes.  Bit-equal to one context."""
    from slam_amd import _lib
    M = sum(sizes)
    X, noise, r = _inputs(M)
    ref = _setup(_lib.F32)
    z_meas = ref.raycast(np.array([3.0, 4.5, 0.0]), want_hits=False)[0][0]
    ref.set_particles(X)
    ref.shard_sample(1.5, 0.0, 0.1, ALPHA, z_meas, None if devnoise else noise, 99, _lib.W_LOG)
    ref.normalise(_lib.W_LOG)
    idx = ref.resample(r, _lib.RESAMPLE_PARALLEL)
    Xr, wr, _ = ref.get_particles()
    res = run_ranks(_rank_fn, len(sizes), sizes, _lib.F32, _lib.W_LOG, z_meas, devnoise, True)
    assert np.array_equal(np.concatenate([x["idx"] for x in res]), idx)
    assert np.array_equal(np.concatenate([x["X"] for x in res]), Xr)
    assert np.array_equal(np.concatenate([x["w"] for x in res]), wr)


@pytest.mark.parametrize("sizes,dtype,mode,devnoise", [((16384, 8192), 1, 1, False), ((8192, 8192), 1, 0, False),
                                                       ((8192, 11000), 0, 1, True)])
def test_two_ranks_equal_one_context(sizes, dtype, mode, devnoise):
    from slam_amd import _lib
    M = sum(sizes)
    X, noise, r = _inputs(M)
    ref = _setup(dtype)
    z_meas = ref.raycast(np.array([3.0, 4.5, 0.0]), want_hits=False)[0][0]
    ref.set_particles(X)
    # the same pass the ranks run: with log-weights only, float32 state takes the float32 beam mixture, whose
    # log-weights differ in the last digits from those of the pass that also forms the raw product
    ref.shard_sample(1.5, 0.0, 0.1, ALPHA, z_meas, None if devnoise else noise, 99, mode)
    total = ref.normalise(mode)
    idx = ref.resample(r, _lib.RESAMPLE_PARALLEL)
    Xr, wr, _ = ref.get_particles()
    res = run_ranks(_rank_fn, len(sizes), sizes, dtype, mode, z_meas, devnoise)
    assert all(x["total"] == total for x in res)
    assert np.array_equal(np.concatenate([x["idx"] for x in res]), idx)
    assert np.array_equal(np.concatenate([x["X"] for x in res]), Xr)
    assert np.array_equal(np.concatenate([x["w"] for x in res]), wr)


def _rccl_rank_fn(rank, world, dist, M, z_meas, device_path, record_cap=256, direct=False, steps=1):
    from slam_amd import _lib
    from slam_amd.sharded import ShardedFilter
    ShardedFilter.RECORD_CAP = record_cap
    X, noise, r = _inputs(M)
    ctx = _setup(_lib.F64)
    ctx.set_particles(X)
    f = ShardedFilter(ctx, dist, 0, 1, M, weight_mode=_lib.W_LOG, force_collectives=True, device_path=device_path,
                      direct_rccl=direct)
    assert f.coll.backend == "nccl" and f.coll.dev.type == "cuda" and f.device_path == device_path
    assert (f.direct is not None) == bool(direct)
    for _ in range(steps):
        f.step(1.5, 0.0, 0.1, ALPHA, z_meas, r, seed=99, noise_ints=noise, want_idx=True)
    Xo, wo, _ = ctx.get_particles()
    return {"idx": f.last_idx, "X": Xo, "w": wo, "total": f.total, "fallbacks": getattr(f, "fallbacks", 0)}


def test_collectives_on_the_steps_own_stream_on_one_rank():
    """The device-driven step with its collectives called through librccl on the step's own stream (a communicator of
    one rank beside torch's): bit-equal to the fused step, over one step and over three."""
    from slam_amd import _lib
    M = 24576
    X, noise, r = _inputs(M)
    for steps in (1, 3):
        ref = _setup(_lib.F64)
        z_meas = ref.raycast(np.array([3.0, 4.5, 0.0]), want_hits=False)[0][0]
        ref.set_particles(X)
        for _ in range(steps):
            ref.sample(1.5, 0.0, 0.1, ALPHA, z_meas, noise, seed=99)
            ref.normalise(_lib.W_LOG)
            idx = ref.resample(r, _lib.RESAMPLE_PARALLEL)
        Xr, wr, _ = ref.get_particles()
        res = run_ranks(_rccl_rank_fn, 1, M, z_meas, True, 256, True, steps, backend="nccl")[0]
        assert res["fallbacks"] == 0
        assert np.array_equal(res["idx"], idx)
        assert np.array_equal(res["X"], Xr) and np.array_equal(res["w"], wr)


@pytest.mark.parametrize("device_path", [False, True])
def test_rccl_path_with_device_tensors_on_one_rank(device_path):
    """The cross-rank path under the "nccl" backend (RCCL) with a group of one, in a fresh process
    that brings torch's HIP runtime up first (as bench.py does): the collectives run on GPU
    tensors, and packed rows travel between torch's device memory and the library's through raw
    device pointers -- the pieces the gloo tests cannot reach.  Result: bit-equal to the fused step."""
    from slam_amd import _lib
    M = 24576
    X, noise, r = _inputs(M)
    ref = _setup(_lib.F64)
    z_meas = ref.raycast(np.array([3.0, 4.5, 0.0]), want_hits=False)[0][0]
    ref.set_particles(X)
    ref.sample(1.5, 0.0, 0.1, ALPHA, z_meas, noise, seed=99)
    total = ref.normalise(_lib.W_LOG)
    idx = ref.resample(r, _lib.RESAMPLE_PARALLEL)
    Xr, wr, _ = ref.get_particles()
    res = run_ranks(_rccl_rank_fn, 1, M, z_meas, device_path, backend="nccl")[0]
    assert device_path or res["total"] == total
    assert np.array_equal(res["idx"], idx)
    assert np.array_equal(res["X"], Xr) and np.array_equal(res["w"], wr)


def _run_lockstep(torch, gens, settle):
    """Runs the device-step generators of several ranks side by side, serving every exchange they yield with plain
    tensor operations; returns what each generator returned."""
    W = len(gens)
    req = [next(g) for g in gens]
    done = [None] * W
    while any(d is None for d in done):
        settle()
        op = req[0][0]
        assert all(q[0] == op for q in req)
        vals = [None] * W
        if op == "all_reduce_max":
            m = torch.stack([q[1] for q in req]).max(0).values
            for q in req:
                q[1].copy_(m)
        elif op == "all_gather":
            cat = torch.cat([q[2] for q in req])
            for q in req:
                q[1].copy_(cat)
        elif op == "all_gather_read":
            cat = torch.cat([q[2] for q in req])
            vals = [cat.cpu().numpy() for _ in req]
        elif op == "all_to_all_fixed":
            C = req[0][1].shape[0] // W
            for d in range(W):
                for s_ in range(W):
                    req[d][1][s_ * C:(s_ + 1) * C].copy_(req[s_][2][d * C:(d + 1) * C])
        elif op == "mark_late":
            pass
        elif op == "read_late":
            vals = [q[1].cpu().numpy() for q in req]
        elif op == "all_to_all":
            offs = [np.concatenate([[0], np.cumsum(q[2])]).astype(int) for q in req]
            for d in range(W):
                parts = [req[s][1][offs[s][d]:offs[s][d + 1]] for s in range(W)]
                assert [len(p) for p in parts] == [int(c) for c in req[d][3]]
                rows = torch.cat(parts) if sum(len(p) for p in parts) else \
                    torch.empty((1, req[d][1].shape[1]), dtype=req[d][1].dtype, device="cuda")
                vals[d] = rows.contiguous()
        settle()
        for k in range(W):
            try:
                req[k] = gens[k].send(vals[k])
            except StopIteration as e:
                done[k] = e.value
    settle()
    return done


def _lockstep_fn(rank, world, dist, sizes, z_meas, record_cap, flat=False, fixed_rows=None):
    """Several ranks of the device-driven step inside ONE process (one context per rank on the same
    GPU); the exchanges between them are done here with plain tensor operations, so that the
    device kernels that combine the ranks' data see real multi-rank inputs."""
    import torch
    from slam_amd import _lib
    from slam_amd.sharded import ShardedFilter
    torch.cuda.set_device(0)
    M = sum(sizes)
    X, noise, r = _inputs(M)
    W = len(sizes)
    ShardedFilter.RECORD_CAP = record_cap
    if fixed_rows is not None:
        ShardedFilter.FIXED_ROWS = fixed_rows
    filt, gens, req = [], [], []
    for k in range(W):
        first = sum(sizes[:k])
        ctx = _setup(_lib.F32, flat)
        ctx.set_particles(X[first:first + sizes[k]])
        f = ShardedFilter.for_layout(ctx, k, sizes)
        filt.append(f)
        gens.append(f._device_step(1.5, 0.0, 0.1, ALPHA, z_meas, r, 99, noise[first:first + sizes[k]]))

    def settle():
        for f in filt:
            f.b.ctx.sync()
        torch.cuda.synchronize()

    done = _run_lockstep(torch, gens, settle)
    out = []
    for k, f in enumerate(filt):
        Xo, wo, _ = f.b.ctx.get_particles()
        out.append({"ok": done[k], "idx": f.b.ctx.get_indices() if done[k] else None, "X": Xo, "w": wo,
                    "rows_sent": f.rows_sent_total, "rows_off_rank": f.rows_sent_off_rank,
                    "exact_exchanges": f.exact_exchanges})
    return out


@pytest.mark.parametrize("sizes,flat", [((16384, 8192), False), ((8192, 8192, 11000), False), ((8192,) * 8, False),
                                        ((8192, 8192, 11000), True), ((8192,) * 8, True)])
def test_device_driven_step_of_several_ranks_equals_one_context(sizes, flat):
    """The device-side combination of several ranks' chunk sums, record lists and drawn-row counts,
    fed with real multi-rank data (ranks run in lockstep inside one process): bit-equal to the
    fused single-context update.  Only rows that some position draws travel: with peaked weights a
    small fraction of the particle set; with equal weights (flat: a map without obstacles) every
    row is drawn once and stays on its rank, up to the rows at the rank boundaries."""
    from slam_amd import _lib
    M = sum(sizes)
    X, noise, r = _inputs(M)
    ref = _setup(_lib.F32, flat)
    z_meas = ref.raycast(np.array([3.0, 4.5, 0.0]), want_hits=False)[0][0]
    ref.set_particles(X)
    ref.shard_sample(1.5, 0.0, 0.1, ALPHA, z_meas, noise, 99, _lib.W_LOG)   # log-weights only, as the ranks run it
    ref.normalise(_lib.W_LOG)
    idx = ref.resample(r, _lib.RESAMPLE_PARALLEL)
    Xr, wr, _ = ref.get_particles()
    res = run_ranks(_lockstep_fn, 1, sizes, z_meas, 256, flat, backend="nccl")[0]
    assert all(x["ok"] for x in res)
    assert all(x["exact_exchanges"] == 0 for x in res)         # (the default blocks hold what pairs of ranks exchange here)
    assert np.array_equal(np.concatenate([x["idx"] for x in res]), idx)
    sent, off_rank = sum(x["rows_sent"] for x in res), sum(x["rows_off_rank"] for x in res)
    print("%d ranks, %d particles, %s weights: %d rows exchanged (%d between ranks), %d distinct ancestors"
          % (len(sizes), M, "equal" if flat else "peaked", sent, off_rank, len(np.unique(idx))))
    assert len(np.unique(idx)) <= sent <= len(np.unique(idx)) + 2 * len(sizes) * len(sizes)
    if flat:
        assert off_rank <= 4 * len(sizes) and sent >= 0.9 * M       # every on-map row once, on its own rank
    else:
        assert sent < 0.5 * M
    assert np.array_equal(np.concatenate([x["X"] for x in res]), Xr)
    assert np.array_equal(np.concatenate([x["w"] for x in res]), wr)


@pytest.mark.parametrize("fixed_rows,flat", [(0, False), (1, False), (1, True), (3, False)])
def test_device_driven_step_with_other_block_sizes(fixed_rows, flat):
    """The row exchange of the device-driven step in blocks of a fixed size per pair of ranks (the default, exercised by
    the test above), with exact counts read by the host in mid-step (fixed_rows = 0), and with blocks too small for what
    a pair has to exchange: the device says so, nothing is resampled, and the step is finished with exact counts --
    the same particles either way."""
    from slam_amd import _lib
    sizes = (8192, 8192, 11000)
    M = sum(sizes)
    X, noise, r = _inputs(M)
    ref = _setup(_lib.F32, flat)
    z_meas = ref.raycast(np.array([3.0, 4.5, 0.0]), want_hits=False)[0][0]
    ref.set_particles(X)
    ref.shard_sample(1.5, 0.0, 0.1, ALPHA, z_meas, noise, 99, _lib.W_LOG)
    ref.normalise(_lib.W_LOG)
    idx = ref.resample(r, _lib.RESAMPLE_PARALLEL)
    Xr, wr, _ = ref.get_particles()
    res = run_ranks(_lockstep_fn, 1, sizes, z_meas, 256, flat, fixed_rows, backend="nccl")[0]
    assert all(x["ok"] for x in res)
    assert np.array_equal(np.concatenate([x["idx"] for x in res]), idx)
    assert np.array_equal(np.concatenate([x["X"] for x in res]), Xr)
    assert np.array_equal(np.concatenate([x["w"] for x in res]), wr)
    off_rank = sum(x["rows_off_rank"] for x in res)
    exact = [x["exact_exchanges"] for x in res]
    print("blocks of %d rows, %s weights: %d rows between ranks, steps finished with exact counts %s"
          % (fixed_rows, "equal" if flat else "peaked", off_rank, exact))
    assert len(set(exact)) == 1                                # the flag is the same on every rank
    if fixed_rows == 0:
        assert exact[0] == 0


def test_device_driven_step_reports_record_overflow():
    from slam_amd import _lib
    sizes = (8192, 8192)
    ref = _setup(_lib.F32)
    z_meas = ref.raycast(np.array([3.0, 4.5, 0.0]), want_hits=False)[0][0]
    res = run_ranks(_lockstep_fn, 1, sizes, z_meas, 1, backend="nccl")[0]
    assert all(x["ok"] is False for x in res)


def test_device_driven_step_falls_back_to_the_host_path_on_overflow():
    """With room for a single record per rank the fixed exchange overflows (24576 weights cross a few
    binades): the step must finish on the host path with the same result."""
    from slam_amd import _lib
    M = 24576
    X, noise, r = _inputs(M)
    ref = _setup(_lib.F64)
    z_meas = ref.raycast(np.array([3.0, 4.5, 0.0]), want_hits=False)[0][0]
    ref.set_particles(X)
    ref.sample(1.5, 0.0, 0.1, ALPHA, z_meas, noise, seed=99)
    total = ref.normalise(_lib.W_LOG)
    idx = ref.resample(r, _lib.RESAMPLE_PARALLEL)
    Xr, wr, _ = ref.get_particles()
    res = run_ranks(_rccl_rank_fn, 1, M, z_meas, True, 1, backend="nccl")[0]
    assert res["fallbacks"] == 1 and res["total"] == total
    assert np.array_equal(res["idx"], idx)
    assert np.array_equal(res["X"], Xr) and np.array_equal(res["w"], wr)


# ---- FastSLAM over several ranks ------------------------------------------------------------
def _fs_inputs(M, steps):
    from oracle import fastslam_oracle as F
    rng = np.random.RandomState(11)
    lm = F.LANDMARKS
    x_gt = [0, 0, 0]
    out = []
    for t in range(steps):
        ut = [25, 0.001] if t != 2 else [25, -np.pi / 2 * 5]
        x_gt = F.true_pose_step(x_gt, ut)
        tm = F.true_measurements(x_gt, lm)
        tr = np.array([float(a[0]) for a in tm]); tp = np.array([float(a[1]) for a in tm])
        tv = np.array([a[2] for a in tm], np.uint8)
        nv = int(tv.sum())
        out.append((ut, tr, tp, tv, rng.randint(-1000, 1000, size=(M, 3, 12)).astype(np.int16), rng.randn(M, 2 * nv),
                    rng.rand() / M))
    return out


def _fs_rank_fn(rank, world, dist, sizes, steps, device_path=None):
    from oracle import fastslam_oracle as F
    from slam_amd import _lib
    from slam_amd.sharded import ShardedFastSLAM
    if device_path:
        import torch
        torch.cuda.set_device(0)
    M = sum(sizes)
    first = sum(sizes[:rank])
    sl = slice(first, first + sizes[rank])
    ctx = _lib.Context(0, _lib.F64)
    f = ShardedFastSLAM(ctx, dist, rank, world, sizes[rank], 16, device_path=device_path)
    assert f.device_path == bool(device_path)
    idx = []
    for (ut, tr, tp, tv, ints, nrm, r) in _fs_inputs(M, steps):
        f.step(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, ints[sl], nrm[sl], r, want_idx=True)
        idx.append(f.last_idx)
    st = ctx.fs_get_state()
    st["idx"] = idx
    return st


@pytest.mark.parametrize("sizes,device_path", [((8192, 8192), None), ((8192, 8192, 3000), None), ((8192, 8192, 3000), True)])
def test_fastslam_over_ranks_equals_one_context(sizes, device_path):
    """FastSLAM with the particles split over ranks (gloo carries the exchanges, the ranks share the
    box's GPU): ancestors and the whole particle state, feature slots included, equal the
    single-context filter bit for bit after four steps."""
    from oracle import fastslam_oracle as F
    from slam_amd import _lib
    M, steps = sum(sizes), 4
    ref = _lib.Context(0, _lib.F64)
    ref.fs_init(M, 16)
    ref_idx = []
    for (ut, tr, tp, tv, ints, nrm, r) in _fs_inputs(M, steps):
        ref_idx.append(ref.fs_step(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, ints, nrm, r, want_idx=True))
    want = ref.fs_get_state()
    res = run_ranks(_fs_rank_fn, len(sizes), sizes, steps, device_path)
    for t in range(steps):
        assert np.array_equal(np.concatenate([x["idx"][t] for x in res]), ref_idx[t]), t
    for k in ("pose", "w", "nfeat", "new_id", "f_m", "f_cov", "f_i", "f_w", "f_id"):
        assert np.array_equal(np.concatenate([x[k] for x in res]), want[k]), k


def _fs_lockstep_fn(rank, world, dist, sizes, steps):
    """Several ranks of the device-driven FastSLAM step inside one process, one context per rank on the same GPU."""
    import torch
    from oracle import fastslam_oracle as F
    from slam_amd import _lib
    from slam_amd.sharded import ShardedFastSLAM
    torch.cuda.set_device(0)
    M, W = sum(sizes), len(sizes)
    filt = [ShardedFastSLAM.for_layout(_lib.Context(0, _lib.F64), k, sizes, 16) for k in range(W)]

    def settle():
        for f in filt:
            f.ctx.sync()
        torch.cuda.synchronize()

    idx = []
    for (ut, tr, tp, tv, ints, nrm, r) in _fs_inputs(M, steps):
        gens = []
        for k, f in enumerate(filt):
            sl = slice(f.firsts[k], f.firsts[k + 1])
            gens.append(f._device_step(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, ints[sl], nrm[sl], r, 0))
        assert all(_run_lockstep(torch, gens, settle))
        idx.append(np.concatenate([f.ctx.get_indices() for f in filt]))
    out = [f.ctx.fs_get_state() for f in filt]
    return {"idx": idx, "state": out, "rows": [(f.rows_sent_total, f.rows_sent_off_rank) for f in filt]}


@pytest.mark.parametrize("sizes", [(8192, 8192), (8192, 8192, 3000), (8192,) * 5])
def test_device_driven_fastslam_of_several_ranks_equals_one_context(sizes):
    """The device-driven FastSLAM step (no host in the loop but for one read of the gathered plan), ranks in lockstep
    inside one process: ancestors and the whole state, feature slots included, equal the single-context filter bit for
    bit after four steps; only drawn particles travel."""
    from oracle import fastslam_oracle as F
    from slam_amd import _lib
    M, steps = sum(sizes), 4
    ref = _lib.Context(0, _lib.F64)
    ref.fs_init(M, 16)
    ref_idx = []
    for (ut, tr, tp, tv, ints, nrm, r) in _fs_inputs(M, steps):
        ref_idx.append(ref.fs_step(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, ints, nrm, r, want_idx=True))
    want = ref.fs_get_state()
    res = run_ranks(_fs_lockstep_fn, 1, sizes, steps, backend="nccl")[0]
    for t in range(steps):
        assert np.array_equal(res["idx"][t], ref_idx[t]), t
    for k in ("pose", "w", "nfeat", "new_id", "f_m", "f_cov", "f_i", "f_w", "f_id"):
        assert np.array_equal(np.concatenate([x[k] for x in res["state"]]), want[k]), k
    sent = sum(a for a, _ in res["rows"])
    print("%d ranks, %d particles, %d steps: %d rows exchanged (%d between ranks)" % (len(sizes), M, steps, sent, sum(b for _, b in res["rows"])))
    assert sent <= steps * (M + len(sizes) ** 2)


def _fs_overflow_fn(rank, world, dist, sizes):
    """One device-driven FastSLAM step of several ranks with room for a single irregular record per rank."""
    import torch
    from oracle import fastslam_oracle as F
    from slam_amd import _lib
    from slam_amd.sharded import ShardedFastSLAM
    torch.cuda.set_device(0)
    ShardedFastSLAM.RECORD_CAP = 1
    M, W = sum(sizes), len(sizes)
    filt = [ShardedFastSLAM.for_layout(_lib.Context(0, _lib.F64), k, sizes, 16) for k in range(W)]

    def settle():
        for f in filt:
            f.ctx.sync()
        torch.cuda.synchronize()

    (ut, tr, tp, tv, ints, nrm, r) = _fs_inputs(M, 1)[0]
    gens = []
    for k, f in enumerate(filt):
        sl = slice(f.firsts[k], f.firsts[k + 1])
        f.t_send.fill_(-7.0)
        gens.append(f._device_step(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, ints[sl], nrm[sl], r, 0))
    raised = ""
    try:
        _run_lockstep(torch, gens, settle)
    except _lib.MclHipError as e:
        raised = str(e)
    settle()
    return {"raised": raised, "planned": [f.t_plan[:W + 1].cpu().numpy().tolist() for f in filt],
            "send_untouched": [bool((f.t_send == -7.0).all().item()) for f in filt]}


def test_device_driven_fastslam_plans_nothing_after_a_record_overflow():
    """After a record merge that ran out of room the running sums are not the exact ones and the draw ends need not
    ascend: the plan must then hold no row at all (nothing is written to the fixed-size send buffer, no particle is
    updated) and carry the overflow word, which the driver turns into an error."""
    res = run_ranks(_fs_overflow_fn, 1, (8192, 8192, 3000), backend="nccl")[0]
    assert "RECORD_CAP" in res["raised"], res["raised"]
    for plan in res["planned"]:
        assert plan[:-1] == [0, 0, 0] and plan[-1] == 1, plan
    assert all(res["send_untouched"])


def _fs_rccl_rank_fn(rank, world, dist, M, steps):
    import torch
    from oracle import fastslam_oracle as F
    from slam_amd import _lib
    from slam_amd.sharded import ShardedFastSLAM
    torch.cuda.set_device(0)
    ctx = _lib.Context(0, _lib.F64)
    f = ShardedFastSLAM(ctx, dist, rank, world, M, 16)
    assert f.device_path
    idx = []
    for (ut, tr, tp, tv, ints, nrm, r) in _fs_inputs(M, steps):
        f.step(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, ints, nrm, r, want_idx=True)
        idx.append(f.last_idx)
    st = ctx.fs_get_state()
    st["idx"] = idx
    return st


def test_device_driven_fastslam_under_rccl_on_one_rank():
    """The same step through torch.distributed on device tensors (RCCL, a group of one on the box's GPU)."""
    from oracle import fastslam_oracle as F
    from slam_amd import _lib
    M, steps = 8192 + 3000, 4
    ref = _lib.Context(0, _lib.F64)
    ref.fs_init(M, 16)
    ref_idx = [ref.fs_step(ut[0], ut[1], F.DT, F.ALPHA, tr, tp, tv, ints, nrm, r, want_idx=True)
               for (ut, tr, tp, tv, ints, nrm, r) in _fs_inputs(M, steps)]
    want = ref.fs_get_state()
    res = run_ranks(_fs_rccl_rank_fn, 1, M, steps, backend="nccl")[0]
    for t in range(steps):
        assert np.array_equal(res["idx"][t], ref_idx[t]), t
    for k in ("pose", "w", "nfeat", "new_id", "f_m", "f_cov", "f_i", "f_w", "f_id"):
        assert np.array_equal(res[k], want[k]), k
