"""Helpers for the multi-rank tests: a numpy stand-in for the per-rank device operations and a
process launcher (gloo, 127.0.0.1)."""
import os
import socket
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

REC = np.dtype([("i", "<i8"), ("w", "<f8"), ("P", "<u8")])


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def synthetic_logw(X):
    """A deterministic stand-in for the beam likelihood: any function of the pose will do for
    testing the cross-rank bookkeeping."""
    return -40.0 * ((X[:, 0] - 3.0) ** 2 + (X[:, 1] - 4.0) ** 2) - 3.0 * np.cos(X[:, 2]) ** 2


class NumpyShardBackend:
    """Per-rank operations in numpy.  Every element is declared an irregular record, which is a
    valid (if slow) classification: the running sum is then rebuilt by the plain sequential
    cumulative sum over the global record list."""

    def __init__(self, X_local):
        self.X = np.array(X_local, np.float64)
        self.w = np.ones(len(self.X))
        self.first, self.m_total = 0, len(self.X)

    def set_range(self, first_index, m_total):
        self.first, self.m_total = first_index, m_total

    def num_local(self):
        return len(self.X)

    def sample(self, v, omega, dt, alpha, z_meas, noise_ints, seed, mode):
        from oracle import mcl_oracle as O
        grid = np.full((200, 200), 255, np.uint8)
        self.X, pm = O.motion(self.X, (v, omega), dt, grid, noise_ints, tuple(alpha))
        self.logw = synthetic_logw(self.X)
        self.w = np.exp(self.logw) * pm

    def max_logw(self):
        return float(np.max(self.logw))

    def weights(self, mode, gmax):
        if mode == 1:
            self.w = np.exp(self.logw - gmax)
        return np.array([np.sum(self.w[s:s + 8192]) for s in range(0, len(self.w), 8192)])

    def scan(self, total, approx_offset):
        self.w = self.w / total
        return 0, len(self.w)

    def get_records(self, n):
        r = np.zeros(n, REC)
        r["i"] = self.first + np.arange(n)
        r["w"] = self.w
        return r

    def finalize(self, all_records, p_off, f_off):
        c = np.cumsum(all_records["w"])
        self.cdf = c[f_off:f_off + len(self.w)]

    def plan_drawn(self, firsts, rank, r):
        """Which local rows some position draws, and the positions of which rank: row j is drawn by the positions
        [G[j-1], G[j]), G[j] = first position m with U_m = r + m/M above the row's running sum (the walk's own
        comparison); a rank's first row always counts as drawn (its predecessor's running sum lives elsewhere), the
        globally last row keeps every position to the end."""
        M, n, world = self.m_total, len(self.cdf), len(firsts) - 1
        U = r + (np.arange(M) / M)
        G = np.searchsorted(U, self.cdf, side="right").astype(np.int64)
        if rank == world - 1:
            G[-1] = M
        drawn = np.ones(n, bool)
        drawn[1:] = G[1:] > G[:-1]
        self._sel = []
        for d in range(world):
            jl = int(np.searchsorted(G, firsts[d], side="right"))
            jh = min(int(np.searchsorted(G, firsts[d + 1] - 1, side="right")), n - 1)
            rows = [j for j in range(jl, jh + 1) if drawn[j]] if firsts[d + 1] > firsts[d] else []
            self._sel.append(np.array(rows, np.int64))
        return np.array([len(x) for x in self._sel], np.int64)

    def pack_drawn(self, world, rows):
        sel = np.concatenate(self._sel)
        gid = (self.first + sel).astype(np.int64).view(np.float64)
        out = np.concatenate([self.X[sel], self.w[sel, None], self.cdf[sel, None], gid[:, None]], 1)
        rows[:len(out)] = __import__("torch").from_numpy(out)

    def unpack_drawn(self, rows, L, r, want_idx):
        R = rows[:L].numpy()
        m = self.first + np.arange(len(self.X))
        U = r + (m / self.m_total)
        j = np.minimum(np.searchsorted(R[:, 4], U, side="left"), L - 1)
        self.X = R[j, :3].copy()
        self.w = R[j, 3].copy()
        return np.ascontiguousarray(R[j, 5]).view(np.int64)


def run_ranks(fn, world, *args, backend="gloo"):
    """fn(rank, world, dist, *args) in `world` fresh processes; returns the per-rank results.  A rendezvous port that
    somebody else took between its choice and the store's listen() (EADDRINUSE) gets one more go with another port."""
    try:
        return _run_ranks_once(fn, world, *args, backend=backend)
    except AssertionError as e:
        if "EADDRINUSE" not in str(e):
            raise
    return _run_ranks_once(fn, world, *args, backend=backend)


def _run_ranks_once(fn, world, *args, backend="gloo"):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_entry, args=(backend, fn, r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    out = {}
    try:
        for _ in range(world):
            r, val = q.get(timeout=300)
            out[r] = val
            if isinstance(val, str) and val.startswith("ERROR"):
                break              # the other ranks may be waiting for this one in a collective: do not wait for them
        for p in procs:
            p.join(60 if len(out) == world else 1)
    finally:
        for p in procs:            # no rank outlives the test (a leftover one would keep holding the GPU)
            if p.is_alive():
                p.terminate()
                p.join(10)
                if p.is_alive():
                    p.kill()
                    p.join(10)
    for r in sorted(out):
        if isinstance(out[r], str) and out[r].startswith("ERROR"):
            raise AssertionError("rank %d failed:\n%s" % (r, out[r]))
    if len(out) != world:
        raise AssertionError("only %d of %d ranks reported" % (len(out), world))
    return [out[r] for r in range(world)]


def _entry(backend, fn, rank, world, port, q, *args):
    try:
        import torch
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        if backend == "nccl":    # torch brings up its HIP runtime before the library is loaded, as bench.py does
            torch.cuda.set_device(rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        val = fn(rank, world, dist, *args)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, val))
    except Exception:
        q.put((rank, "ERROR " + traceback.format_exc()))
