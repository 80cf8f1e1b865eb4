"""One filter over several GPUs: one process per GPU, contiguous particle ranges per rank, map
and beam table replicated.

Only small per-rank results cross ranks as collectives -- the largest log-weight (all-reduce
max), the sums of the 8192-element chunks (all-gather; every rank then adds them left to right,
so the global sum is bit-equal to np.sum of the whole vector on every rank), the integer totals
and the short irregular-record lists of the order-exact scan (all-gather), and the number of rows
every rank sends to every other (all-gather) -- followed by one all-to-all of rows.  Only rows that
some position draws travel (mc_localization.py:253-259 reads X[i] only for the drawn i): a rank
finds, from its exact running sums, the range of positions that draw each of its rows.

The arithmetic of every step lives behind a `backend` (the C ABI in production); this file is
the cross-rank bookkeeping, and it is the same code under RCCL (GPU tensors) and gloo (CPU
tensors, used by the CPU tests and by multi-process tests that share one GPU).
"""
import numpy as np

from . import _lib

CHUNK = 8192


class HipShardBackend:
    """The per-rank operations on a `_lib.Context`."""

    def __init__(self, ctx):
        self.ctx = ctx

    def set_range(self, first_index, m_total):
        self.ctx.shard_set(first_index, m_total)

    def num_local(self):
        return self.ctx.num_particles()

    def sample(self, v, omega, dt, alpha, z_meas, noise_ints, seed, mode):
        self.ctx.shard_sample(v, omega, dt, alpha, z_meas, noise_ints, seed, mode)

    def max_logw(self):
        return self.ctx.shard_max_logw()

    def weights(self, mode, gmax):
        return self.ctx.shard_weights(mode, gmax)

    def scan(self, total, approx_offset):
        return self.ctx.shard_scan(total, approx_offset)

    def get_records(self, n):
        return self.ctx.shard_get_records(n)

    def finalize(self, all_records, p_off, f_off):
        self.ctx.shard_finalize(all_records, p_off, f_off)

    def plan_drawn(self, firsts, rank, r):
        """Number of drawn local rows every rank needs."""
        return self.ctx.shard_plan_drawn(firsts, rank, r)

    def pack_drawn(self, world, rows):
        """rows: torch tensor [n,6] float64 (GPU or CPU) that receives the drawn rows, destination after destination."""
        self.ctx.shard_pack_drawn(world, rows.data_ptr())

    def unpack_drawn(self, rows, L, r, want_idx):
        return self.ctx.shard_unpack_drawn(rows.data_ptr(), L, r, want_idx)

    def step_single(self, v, omega, dt, alpha, z_meas, r, noise_ints, seed, mode, wait):
        self.ctx.step(v, omega, dt, alpha, z_meas, r, noise_ints, seed, mode, wait=wait)


class _Coll:
    """The few collectives the filter needs, on GPU tensors under RCCL or CPU tensors under gloo."""

    def __init__(self, dist, rank, world):
        import torch
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        self.backend = dist.get_backend()
        self.dev = torch.device("cuda", torch.cuda.current_device()) if self.backend == "nccl" else torch.device("cpu")

    def _t(self, a, dtype):
        return self.torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.dev)

    def allreduce_max(self, x):
        t = self._t([x], self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.cpu()[0])

    def allreduce_sum_i64(self, a):
        t = self._t(a, self.torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def allgather_bytes(self, payload, cap=None):
        """Every rank's byte string, in rank order (lengths may differ).  With `cap` (the same on
        every rank) one collective does it as long as every payload fits in cap bytes: the length
        travels in front of the payload; otherwise lengths are exchanged first."""
        torch = self.torch
        if cap is not None:
            buf = np.zeros(8 + cap, np.uint8)
            buf[:8] = np.frombuffer(np.int64(len(payload)).tobytes(), np.uint8)
            fits = len(payload) <= cap
            if fits:
                buf[8:8 + len(payload)] = np.frombuffer(payload, np.uint8)
            t = self._t(buf, torch.uint8)
            outs = [torch.zeros_like(t) for _ in range(self.world)]
            self.dist.all_gather(outs, t)
            outs = [o.cpu().numpy() for o in outs]
            lens = [int(np.frombuffer(o[:8].tobytes(), np.int64)[0]) for o in outs]
            if all(n <= cap for n in lens):
                return [o[8:8 + lens[i]].tobytes() for i, o in enumerate(outs)]
            # some rank did not fit (every rank sees that): fall through to the two-step exchange
        n = self._t([len(payload)], torch.int64)
        lens = [torch.zeros_like(n) for _ in range(self.world)]
        self.dist.all_gather(lens, n)
        lens = [int(x.cpu()[0]) for x in lens]
        size = max(max(lens), 1)
        buf = np.zeros(size, np.uint8)
        buf[:len(payload)] = np.frombuffer(payload, np.uint8)
        t = self._t(buf, torch.uint8)
        outs = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return [o.cpu().numpy()[:lens[i]].tobytes() for i, o in enumerate(outs)]

    def rows(self, n, width=5, dtype=None):
        return self.torch.empty((max(n, 1), width), dtype=dtype or self.torch.float64, device=self.dev)

    def all_to_all_rows(self, send, send_counts, recv_counts):
        """Row slices to every rank (send_counts[d] rows to rank d, in order)."""
        torch, dist = self.torch, self.dist
        recv = self.rows(int(sum(recv_counts)), send.shape[1], send.dtype)
        if self.backend == "nccl":
            dist.all_to_all_single(recv[:int(sum(recv_counts))], send[:int(sum(send_counts))],
                                   [int(c) for c in recv_counts], [int(c) for c in send_counts])
            torch.cuda.synchronize()
            return recv
        so = np.concatenate([[0], np.cumsum(send_counts)]).astype(int)
        ro = np.concatenate([[0], np.cumsum(recv_counts)]).astype(int)
        reqs = []
        for peer in range(self.world):
            if peer == self.rank:
                if send_counts[peer]:
                    recv[ro[peer]:ro[peer + 1]] = send[so[peer]:so[peer + 1]]
                continue
            if send_counts[peer]:
                reqs.append(dist.isend(send[so[peer]:so[peer + 1]].contiguous(), peer))
            if recv_counts[peer]:
                reqs.append(dist.irecv(recv[ro[peer]:ro[peer + 1]], peer))
        for q in reqs:
            q.wait()
        return recv


def sequential_sum(values):
    """Left-to-right float64 sum starting from 0.0 -- the order in which numpy adds chunk results."""
    t = np.float64(0.0)
    for v in np.asarray(values, np.float64):
        t = t + v
    return float(t)


def drive_exchanges(gen, torch, dist, stream, clock=None, direct=None):
    """Serve the exchanges a device-step generator asks for with torch.distributed, on the step's stream.  Under RCCL
    the collectives read and write the device tensors; under gloo (ranks that share one GPU, in tests) the same calls
    run on host copies.  clock: a dict that receives the host's time per step -- "enqueue_s" up to the step's one wait,
    "wait_s" in it -- for reports.  direct: a rccl_direct.DirectRccl, whose collectives are entries of the step's own stream
    (the all-reduce, the all-gathers and the all-to-all of equal blocks; what else a step may need stays with torch)."""
    import time
    staged = dist.get_backend() != "nccl"
    if staged:
        direct = None
    raw_stream = stream.cuda_stream if (direct is not None and stream is not None) else None
    t_begin = time.perf_counter()
    late_event = None

    def gather(out, inp):
        if direct is not None:
            direct.all_gather(out, inp, raw_stream)
        elif staged:
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host, inp.cpu())
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, inp)

    with torch.cuda.stream(stream):
        try:
            req = next(gen)
            while True:
                op = req[0]
                val = None
                if op == "all_reduce_max":
                    if direct is not None:
                        direct.all_reduce_max_f64(req[1], raw_stream)
                    elif staged:
                        host = req[1].cpu()
                        dist.all_reduce(host, op=dist.ReduceOp.MAX)
                        req[1].copy_(host)
                    else:
                        dist.all_reduce(req[1], op=dist.ReduceOp.MAX)
                elif op == "all_gather":
                    gather(req[1], req[2])
                elif op == "all_gather_read":
                    gather(req[1], req[2])
                    val = req[1].cpu().numpy()
                elif op == "all_to_all":
                    send, sc, rc = req[1], req[2], req[3]
                    n_in, n_out = int(sum(sc)), int(sum(rc))
                    val = torch.empty((max(n_out, 1), send.shape[1]), dtype=send.dtype, device=send.device)
                    if staged:
                        host = torch.empty((n_out, send.shape[1]), dtype=send.dtype)
                        dist.all_to_all_single(host, send[:n_in].cpu(), [int(c) for c in rc], [int(c) for c in sc])
                        val[:n_out].copy_(host)
                    else:
                        dist.all_to_all_single(val[:n_out], send[:n_in], [int(c) for c in rc], [int(c) for c in sc])
                elif op == "all_to_all_fixed":
                    recv, send = req[1], req[2]            # equal blocks, one per rank
                    if direct is not None:
                        direct.all_to_all_equal(recv, send, raw_stream)
                    elif staged:
                        host = torch.empty(recv.shape, dtype=recv.dtype)
                        dist.all_to_all_single(host, send.cpu())
                        recv.copy_(host)
                    else:
                        dist.all_to_all_single(recv, send)
                elif op == "mark_late":
                    # what the host will read at the end of the step has been stored into its page-locked memory by the
                    # kernel just enqueued: an event behind that kernel, so that the wait below ends while the step's
                    # last kernels still run and the host has the next step's first kernels enqueued before the GPU
                    # falls idle
                    late_event = None
                    if len(req) > 2 and req[2] is not None:
                        late_event = torch.cuda.Event()
                        late_event.record()
                elif op == "read_late":
                    t_wait = time.perf_counter()
                    # everything of the step is enqueued: the one wait, at its end (into page-locked memory where there is some)
                    if len(req) > 2 and req[2] is not None and late_event is not None:
                        late_event.synchronize()
                        val = req[2].numpy().copy()
                    else:
                        val = req[1].cpu().numpy()
                    if clock is not None:
                        clock["enqueue_s"] = clock.get("enqueue_s", 0.0) + (t_wait - t_begin)
                        clock["wait_s"] = clock.get("wait_s", 0.0) + (time.perf_counter() - t_wait)
                req = gen.send(val)
        except StopIteration as done:
            return done.value


class ShardedFilter:
    """Drives one MCL update per call over `world` ranks."""

    RECORD_CAP = 256     # irregular records per rank that fit the fixed-size exchange of the device-driven step
    # Rows per pair of ranks in the device-driven step's row exchange (0: exact counts, read by the host in the middle
    # of the step).  With blocks of a fixed size nothing the host needs to know stands between the first kernel of a
    # step and its last: the host enqueues the whole step while the sector query runs, and looks at one flag at the
    # end.  Rows a rank draws from itself do not count (they never enter the exchange); between two ranks a handful
    # of rows is the rule, a pair that has more finishes that step with exact counts.
    FIXED_ROWS = 2048

    def __init__(self, ctx_or_backend, dist, rank, world, m_local, weight_mode=_lib.W_LOG, force_collectives=False,
                 device_path=None, direct_rccl=None):
        self.b = HipShardBackend(ctx_or_backend) if isinstance(ctx_or_backend, _lib.Context) else ctx_or_backend
        self.dist, self.rank, self.world, self.mode = dist, rank, world, weight_mode
        self.m_local = int(m_local)
        self.last_idx = None
        self.collective = world > 1 or force_collectives   # a single rank may run the cross-rank path too (tests)
        if self.collective:
            self.coll = _Coll(dist, rank, world)
            sizes = self.coll.allgather_bytes(np.int64(self.m_local).tobytes())
            self.sizes = [int(np.frombuffer(s, np.int64)[0]) for s in sizes]
            for s in self.sizes[:-1]:
                if s % CHUNK:
                    raise ValueError("every rank but the last must hold a multiple of %d particles" % CHUNK)
            self.firsts = [int(x) for x in np.concatenate([[0], np.cumsum(self.sizes)])]
            self.m_total = self.firsts[-1]
            self.b.set_range(self.firsts[rank], self.m_total)
            self.chunk_cap = 8 * max((n + CHUNK - 1) // CHUNK for n in self.sizes)   # bytes of the longest chunk-sum list
            if device_path is None:
                device_path = (self.coll.backend == "nccl" and weight_mode == _lib.W_LOG and
                               isinstance(self.b, HipShardBackend))
            self.device_path = bool(device_path)
            self.direct = None
            if self.device_path:
                torch = self.coll.torch      # (the step's tensors live on the GPU whatever carries the collectives)
                self._device_setup(torch, torch.device("cuda", torch.cuda.current_device()))
                if direct_rccl is None:
                    import os
                    direct_rccl = os.environ.get("MCLHIP_RCCL_DIRECT", "0") == "1"
                if direct_rccl and self.coll.backend == "nccl" and self.stream is not None:
                    from .rccl_direct import DirectRccl
                    self.direct = DirectRccl(torch, dist, rank, world)   # (collectives as entries of the step's own stream)
        else:
            self.sizes, self.firsts, self.m_total = [self.m_local], [0, self.m_local], self.m_local
            self.device_path = False
            self.direct = None

    @classmethod
    def for_layout(cls, ctx, rank, sizes, weight_mode=_lib.W_LOG):
        """A device-path filter for rank `rank` of a known layout, without a process group: whoever
        drives its _device_step generator supplies the exchanges (used to run several ranks inside
        one process)."""
        import torch
        self = cls.__new__(cls)
        self.b = HipShardBackend(ctx)
        self.dist, self.rank, self.world, self.mode = None, rank, len(sizes), weight_mode
        self.sizes = [int(n) for n in sizes]
        self.m_local = self.sizes[rank]
        self.firsts = [int(x) for x in np.concatenate([[0], np.cumsum(self.sizes)])]
        self.m_total = self.firsts[-1]
        self.last_idx, self.collective, self.device_path, self.coll = None, True, True, None
        self.direct = None
        self.b.set_range(self.firsts[rank], self.m_total)
        self.chunk_cap = 8 * max((n + CHUNK - 1) // CHUNK for n in self.sizes)
        self._device_setup(torch, torch.device("cuda", torch.cuda.current_device()))
        return self

    # -- device-driven step: collectives on device tensors, one host read per step ------------
    def _device_setup(self, torch, dev):
        W = self.world
        cmax = self.chunk_cap // 8
        rb = 2 + 3 * self.RECORD_CAP          # 64-bit words per rank: header + records of 24 bytes
        self.stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self.t_max = torch.zeros(1, dtype=torch.float64, device=dev)
        self.t_chunks = torch.zeros(cmax, dtype=torch.float64, device=dev)
        self.t_all_chunks = torch.zeros(W * cmax, dtype=torch.float64, device=dev)
        self.t_rec = torch.zeros(rb, dtype=torch.int64, device=dev)
        self.t_all_rec = torch.zeros(W * rb, dtype=torch.int64, device=dev)
        self.t_counts = torch.zeros(W + 1, dtype=torch.int64, device=dev)            # rows for every rank + overflow flag
        self.t_late = torch.zeros(W * (W + 1) + 1, dtype=torch.int64, device=dev)    # what the host reads: the gathered counts ...
        self.t_all_counts = self.t_late[:W * (W + 1)]
        self.t_flag = self.t_late[W * (W + 1):]                                      # ... and the fixed exchange's flag
        self.h_late = None
        if dev.type == "cuda":
            try:
                self.h_late = torch.empty(W * (W + 1) + 1, dtype=torch.int64).pin_memory()
            except RuntimeError:
                self.h_late = None
        self.fixed_rows = int(self.FIXED_ROWS)
        if self.fixed_rows:
            self.t_blocks_send = torch.zeros((W * self.fixed_rows, 6), dtype=torch.float64, device=dev)
            self.t_blocks_recv = torch.zeros((W * self.fixed_rows, 6), dtype=torch.float64, device=dev)
        self.exact_exchanges = 0           # steps whose rows did not fit the fixed blocks
        self.clock = {}                    # host seconds: enqueuing the steps, waiting at their ends
        self.rows_sent_off_rank = 0        # running totals, for reports
        self.rows_sent_total = 0
        self.steps_done = 0
        self._torch = torch
        self._keep = None
        if self.stream is not None:
            self.b.ctx.set_stream(self.stream.cuda_stream)

    def _device_step(self, v, omega, dt, alpha, z_meas, r, seed, noise_ints):
        """Generator: enqueues this rank's work and yields the exchanges it needs, in order --
        ("all_reduce_max", t), ("all_gather", out, inp), ("all_gather_read", out, inp) -> numpy,
        ("all_to_all", send, send_counts, recv_counts) -> recv, ("all_to_all_fixed", recv, send) with equal blocks,
        ("mark_late", t, pinned) (the kernel before it has stored t into the page-locked copy), ("read_late", t, pinned) -> numpy.  Returns False if a rank had more
        irregular records than the fixed exchange holds (the caller then finishes on the host path)."""
        ctx, rank, world, cap = self.b.ctx, self.rank, self.world, self.RECORD_CAP
        ctx.dshard_sample_max(v, omega, dt, alpha, z_meas, noise_ints, seed, self.mode, self.t_max.data_ptr())
        yield ("all_reduce_max", self.t_max)
        ctx.dshard_weights(self.t_max.data_ptr(), self.t_chunks.data_ptr())
        yield ("all_gather", self.t_all_chunks, self.t_chunks)
        ctx.dshard_scan(self.t_all_chunks.data_ptr(), self.firsts, rank, self.t_rec.data_ptr(), cap)
        yield ("all_gather", self.t_all_rec, self.t_rec)
        ctx.dshard_finalize(self.t_all_rec.data_ptr(), self.firsts, rank, cap, r, self.t_counts.data_ptr())
        if self.fixed_rows:
            # blocks of a fixed size: the gathered counts stay on the device, which also decides whether every block
            # was large enough; the host reads that, and the counts for its report, once the step is enqueued
            yield ("all_gather", self.t_all_counts, self.t_counts)
            ctx.dshard_pack_fixed(world, rank, self.fixed_rows, self.t_blocks_send.data_ptr(),
                                  self.t_all_counts.data_ptr(), self.t_flag.data_ptr(),
                                  self.h_late.data_ptr() if self.h_late is not None else None)
            yield ("mark_late", self.t_late, self.h_late)          # flag and counts are on their way to the host here already
            yield ("all_to_all_fixed", self.t_blocks_recv, self.t_blocks_send)
            ctx.dshard_unpack_fixed(self.t_blocks_recv.data_ptr(), self.t_all_counts.data_ptr(), world, rank,
                                    self.fixed_rows, r, self.t_flag.data_ptr())
            late = yield ("read_late", self.t_late, self.h_late)
            flag = int(late[-1])
            counts = late[:-1].reshape(world, world + 1)
            if flag:
                ctx.dshard_unpack_undo()       # nothing was resampled (the same on every rank)
        else:
            # the one host read of the step: how many rows every rank sends to every other (only rows that some
            # position draws travel; the all-to-all needs its split sizes on the host)
            counts = yield ("all_gather_read", self.t_all_counts, self.t_counts)
            counts = counts.reshape(world, world + 1)
            flag = 2
        if counts[:, -1].any():
            return False
        send_counts = [int(c) for c in counts[rank, :world]]
        recv_counts = [int(c) for c in counts[:world, rank]]
        n_send, L = sum(send_counts), sum(recv_counts)
        self.steps_done += 1
        self.rows_sent_total += n_send
        self.rows_sent_off_rank += n_send - send_counts[rank]
        if flag == 0:
            return True
        self.exact_exchanges += 1 if self.fixed_rows else 0
        send = self._torch.empty((max(n_send, 1), 6), dtype=self._torch.float64, device=self.t_max.device)
        if n_send:
            ctx.dshard_pack(world, send.data_ptr())
        recv = yield ("all_to_all", send, send_counts, recv_counts)
        ctx.dshard_unpack_resample(recv.data_ptr(), L, r)
        self._keep = (send, recv)       # alive until the next step has been enqueued behind them
        return True

    def _drive(self, gen):
        return drive_exchanges(gen, self._torch, self.dist, self.stream, self.clock, self.direct)

    def step(self, v, omega, dt, alpha, z_meas, r, seed=0, noise_ints=None, want_idx=False):
        """r = (uniform draw) / M_total, identical on every rank."""
        if not self.collective:
            # pre-drawn integers are read from the caller's buffer, so that call waits as well
            wait = bool(want_idx) or noise_ints is not None
            self.b.step_single(v, omega, dt, alpha, z_meas, r, noise_ints, seed, self.mode, wait=wait)
            self.last_idx = self.b.ctx.get_indices() if want_idx and isinstance(self.b, HipShardBackend) else None
            return
        if self.device_path:
            ok = self._drive(self._device_step(v, omega, dt, alpha, z_meas, r, seed, noise_ints))
            if ok:
                self.total = None
                self.last_idx = self.b.ctx.get_indices() if want_idx else None
                return
            self.fallbacks = getattr(self, "fallbacks", 0) + 1
            if self.coll is None:
                raise RuntimeError("more irregular records than the fixed exchange holds, and this filter has no "
                                   "process group to finish the step on the host path (built by for_layout)")
            self._finish_on_host(r, want_idx)    # too many irregular records for the fixed exchange
            return
        self.b.sample(v, omega, dt, alpha, z_meas, noise_ints, seed, self.mode)
        self._finish_on_host(r, want_idx)

    def _finish_on_host(self, r, want_idx):
        """Everything after the sampling, with the host combining the ranks' small results."""
        b, co, rank, world = self.b, self.coll, self.rank, self.world
        gmax = co.allreduce_max(b.max_logw()) if self.mode == _lib.W_LOG else 0.0
        # global sum in numpy's order: chunk sums of all ranks, added left to right
        chunks = b.weights(self.mode, gmax)
        all_chunks = [np.frombuffer(x, np.float64) for x in co.allgather_bytes(chunks.tobytes(), cap=self.chunk_cap)]
        total = sequential_sum(np.concatenate(all_chunks))
        self.total = total
        below = float(np.sum(np.concatenate(all_chunks[:rank]))) if rank else 0.0
        with np.errstate(divide="ignore", invalid="ignore"):
            approx_offset = float(np.float64(below) / np.float64(total))
        # local half of the order-exact scan, then the global record list
        k_total, nrec = b.scan(total, approx_offset)
        recs = b.get_records(nrec)
        head = np.array([k_total, nrec], np.uint64).tobytes()
        parts = co.allgather_bytes(head + recs.tobytes(), cap=16 + 24 * 256)   # a few records per rank is the rule
        k_tot = [int(np.frombuffer(p[:16], np.uint64)[0]) for p in parts]
        n_rec = [int(np.frombuffer(p[:16], np.uint64)[1]) for p in parts]
        p_off = [0] * world
        f_off = [0] * world
        for s in range(1, world):
            p_off[s] = (p_off[s - 1] + k_tot[s - 1]) & 0xFFFFFFFFFFFFFFFF
            f_off[s] = f_off[s - 1] + n_rec[s - 1]
        glob = []
        for s, p in enumerate(parts):
            rs = np.frombuffer(p[16:], _lib.Context.RECORD_DTYPE).copy()
            rs["P"] = rs["P"] + np.uint64(p_off[s])  # wraps modulo 2^64 like the device prefix
            glob.append(rs)
        b.finalize(np.concatenate(glob), p_off[rank], f_off[rank])
        # only the rows that some position draws travel: counts per destination, agreed over the ranks
        mine = b.plan_drawn(self.firsts, rank, r)
        counts = np.stack([np.frombuffer(x, np.int64) for x in co.allgather_bytes(np.asarray(mine, np.int64).tobytes())])
        send_counts = [int(c) for c in counts[rank]]          # [source, destination]
        recv_counts = [int(c) for c in counts[:, rank]]
        n_send, L = sum(send_counts), sum(recv_counts)
        self.rows_sent_total = getattr(self, "rows_sent_total", 0) + n_send
        self.rows_sent_off_rank = getattr(self, "rows_sent_off_rank", 0) + n_send - send_counts[rank]
        self.steps_done = getattr(self, "steps_done", 0) + 1
        send = co.rows(n_send, 6)
        if n_send:
            b.pack_drawn(world, send)
        recv = co.all_to_all_rows(send, send_counts, recv_counts)
        self.last_idx = b.unpack_drawn(recv, L, r, want_idx)


class ShardedFastSLAM:
    """One FastSLAM 1.0 filter over `world` ranks (13_fast_slam/fast_slam1.py:240-433): every rank runs
    the per-particle step on its contiguous range, the ranks then agree on the weight sum and the
    running sums in the reference's strictly sequential order, and the particles that some position
    draws travel as pose rows plus feature rows in two all-to-all exchanges.

    Two drivers of the same arithmetic: device-driven (default under RCCL) -- every library call only
    enqueues, the collectives run on device tensors, one device-to-host read per step -- and
    host-combined (gloo; the exchanges are the ones of ShardedFilter's host-combined step)."""

    RECORD_CAP = 4096    # irregular records per rank in the device-driven exchange: the running sum of the raw weights
                         # crosses every binade between the smallest weight and the total at most once on ALL ranks together

    def __init__(self, ctx, dist, rank, world, m_local, capacity, device_path=None):
        self.ctx, self.dist, self.rank, self.world = ctx, dist, rank, world
        ctx.fs_init(int(m_local), int(capacity))
        self.capacity = int(capacity)
        self.coll = _Coll(dist, rank, world)
        sizes = self.coll.allgather_bytes(np.int64(m_local).tobytes())
        self.sizes = [int(np.frombuffer(x, np.int64)[0]) for x in sizes]
        for n in self.sizes[:-1]:
            if n % CHUNK:
                raise ValueError("every rank but the last must hold a multiple of %d particles" % CHUNK)
        self.firsts = [int(x) for x in np.concatenate([[0], np.cumsum(self.sizes)])]
        self.m_total = self.firsts[-1]
        ctx.shard_set(self.firsts[rank], self.m_total)
        self.words = ctx.fs_row_words()
        self.last_idx = None
        if device_path is None:
            device_path = self.coll.backend == "nccl"
        self.device_path = bool(device_path)
        if self.device_path:
            torch = self.coll.torch
            self._device_setup(torch, torch.device("cuda", torch.cuda.current_device()))

    @classmethod
    def for_layout(cls, ctx, rank, sizes, capacity):
        """A device-path filter for rank `rank` of a known layout, without a process group: whoever drives its
        _device_step generator supplies the exchanges (several ranks inside one process, for tests)."""
        import torch
        self = cls.__new__(cls)
        self.ctx, self.dist, self.rank, self.world, self.coll = ctx, None, rank, len(sizes), None
        ctx.fs_init(int(sizes[rank]), int(capacity))
        self.capacity = int(capacity)
        self.sizes = [int(n) for n in sizes]
        self.firsts = [int(x) for x in np.concatenate([[0], np.cumsum(self.sizes)])]
        self.m_total = self.firsts[-1]
        ctx.shard_set(self.firsts[rank], self.m_total)
        self.words = ctx.fs_row_words()
        self.last_idx, self.device_path = None, True
        self._device_setup(torch, torch.device("cuda", torch.cuda.current_device()))
        return self

    # -- device-driven step ---------------------------------------------------------------------
    def _device_setup(self, torch, dev):
        W, m_local = self.world, self.sizes[self.rank]
        cmax = max((n + CHUNK - 1) // CHUNK for n in self.sizes)
        rb = 2 + 3 * self.RECORD_CAP          # 64-bit words per rank: header + records of 24 bytes
        self._torch = torch
        self.stream = torch.cuda.Stream(device=dev)
        f64, i64 = torch.float64, torch.int64
        self.t_chunks = torch.zeros(cmax, dtype=f64, device=dev)
        self.t_all_chunks = torch.zeros(W * cmax, dtype=f64, device=dev)
        self.t_rec = torch.zeros(rb, dtype=i64, device=dev)
        self.t_all_rec = torch.zeros(W * rb, dtype=i64, device=dev)
        self.t_last = torch.zeros(2, dtype=f64, device=dev)                 # last running sum of the raw weights, overflow flag
        self.t_all_last = torch.zeros(2 * W, dtype=f64, device=dev)
        self.t_plan = torch.zeros(W + 1 + 3, dtype=i64, device=dev)         # rows per destination, record overflow; then the select info
        self.t_all_plan = torch.zeros(W * (W + 4), dtype=i64, device=dev)
        self.t_send = torch.empty((m_local + W, 6), dtype=f64, device=dev)  # every local row once, boundary rows twice
        self.rows_sent_total = self.rows_sent_off_rank = self.steps_done = 0
        self._keep = None
        self.ctx.set_stream(self.stream.cuda_stream)

    def _device_step(self, v, omega, dt, alpha, truth_r, truth_phi, truth_valid, ints, normals, r, seed):
        """Generator: enqueues this rank's work and yields the exchanges it needs, in the vocabulary of
        ShardedFilter._device_step.  One host read: the gathered plan (row counts, longest travelling feature list,
        overflow flags)."""
        ctx, rank, world, cap, torch = self.ctx, self.rank, self.world, self.RECORD_CAP, self._torch
        ctx.fs_dshard_kernel(v, omega, dt, alpha, truth_r, truth_phi, truth_valid, ints, normals, seed, self.t_chunks.data_ptr())
        yield ("all_gather", self.t_all_chunks, self.t_chunks)
        # sum_w exactly as `for m in range(M): sum_w += Y[m].w` forms it (:411-413): the last running sum of the raw weights
        ctx.fs_dshard_scan(self.t_all_chunks.data_ptr(), None, self.firsts, rank, self.t_rec.data_ptr(), cap)
        yield ("all_gather", self.t_all_rec, self.t_rec)
        ctx.fs_dshard_finalize_raw(self.t_all_rec.data_ptr(), self.firsts, rank, cap, self.t_last.data_ptr())
        yield ("all_gather", self.t_all_last, self.t_last)
        # w /= max(1e-8, sum_w) (:415-418) and the running sums c_i (:421-428)
        ctx.fs_dshard_scan(self.t_all_chunks.data_ptr(), self.t_all_last.data_ptr(), self.firsts, rank, self.t_rec.data_ptr(), cap)
        yield ("all_gather", self.t_all_rec, self.t_rec)
        ctx.dshard_finalize(self.t_all_rec.data_ptr(), self.firsts, rank, cap, r, self.t_plan.data_ptr())
        # pose rows of the particles some position draws; :318-404 for exactly those
        ctx.fs_dshard_select(world, self.t_plan.data_ptr(), self.t_last.data_ptr(), self.t_send.data_ptr(),
                             self.t_plan.data_ptr() + 8 * (world + 1))
        plan = yield ("all_gather_read", self.t_all_plan, self.t_plan)
        plan = plan.reshape(world, world + 4)
        # Either error leaves the step half done on every rank (poses advanced, no particle resampled): the filter
        # has to be set up again -- with a larger capacity, or a larger RECORD_CAP -- before it is stepped further.
        # (On a record overflow the plan holds no rows: nothing was packed or updated.)
        if plan[:, world].any() or plan[:, world + 3].any():
            raise _lib.MclHipError("more irregular records on a rank than the fixed exchange holds (ShardedFastSLAM.RECORD_CAP)")
        if plan[:, world + 2].any():
            raise _lib.MclHipError("a particle ran out of feature slots: raise the capacity")
        send_counts = [int(c) for c in plan[rank, :world]]
        recv_counts = [int(c) for c in plan[:world, rank]]
        n_send, L = sum(send_counts), sum(recv_counts)
        most = max(1, min(int(plan[:, world + 1].max()), self.capacity))
        self.steps_done += 1
        self.rows_sent_total += n_send
        self.rows_sent_off_rank += n_send - send_counts[rank]
        ctx.fs_set_row_slots(most)           # feature rows as long as the longest list that travels, not as the capacity
        self.words = ctx.fs_row_words()
        fsend = torch.empty((max(n_send, 1), self.words), dtype=torch.int64, device=self.t_send.device)
        ctx.fs_dshard_pack(n_send, fsend.data_ptr())
        recv = yield ("all_to_all", self.t_send, send_counts, recv_counts)
        frecv = yield ("all_to_all", fsend, send_counts, recv_counts)
        ctx.fs_dshard_unpack(recv.data_ptr(), frecv.data_ptr(), L, r, most)
        self._keep = (fsend, recv, frecv)    # alive until the next step has been enqueued behind them
        return True

    def _exact_running_sums(self, divisor, all_chunks):
        """w /= divisor on every rank, then the strictly sequential running sum of the GLOBAL weight
        vector, this rank's part left on the device (the record exchange of the particle filter)."""
        ctx, co, rank, world = self.ctx, self.coll, self.rank, self.world
        below = float(np.sum(np.concatenate(all_chunks[:rank]))) if rank else 0.0
        k_total, nrec = ctx.shard_scan(divisor, below / divisor)
        recs = ctx.shard_get_records(nrec)
        parts = co.allgather_bytes(np.array([k_total, nrec], np.uint64).tobytes() + recs.tobytes(), cap=16 + 24 * 256)
        k_tot = [int(np.frombuffer(p[:16], np.uint64)[0]) for p in parts]
        n_rec = [int(np.frombuffer(p[:16], np.uint64)[1]) for p in parts]
        p_off, f_off = [0] * world, [0] * world
        for s in range(1, world):
            p_off[s] = (p_off[s - 1] + k_tot[s - 1]) & 0xFFFFFFFFFFFFFFFF
            f_off[s] = f_off[s - 1] + n_rec[s - 1]
        glob = []
        for s, p in enumerate(parts):
            rs = np.frombuffer(p[16:], _lib.Context.RECORD_DTYPE).copy()
            rs["P"] = rs["P"] + np.uint64(p_off[s])
            glob.append(rs)
        ctx.shard_finalize(np.concatenate(glob), p_off[rank], f_off[rank])

    def step(self, v, omega, dt, alpha, truth_r, truth_phi, truth_valid, ints, normals, r, seed=0, want_idx=False):
        """ints / normals: this rank's slice of the pre-drawn numbers (or None: device generators keyed
        by the global particle index).  r = np.random.rand() * 1/M_total, identical on every rank."""
        ctx, co, rank, world = self.ctx, self.coll, self.rank, self.world
        if self.device_path:
            drive_exchanges(self._device_step(v, omega, dt, alpha, truth_r, truth_phi, truth_valid, ints, normals, r, seed),
                            self._torch, self.dist, self.stream)
            self.last_idx = ctx.get_indices() if want_idx else None
            return
        ctx.fs_shard_kernel(v, omega, dt, alpha, truth_r, truth_phi, truth_valid, ints, normals, seed)
        chunks = ctx.shard_weights(_lib.W_PRODUCT, 0.0)          # approximate offsets only
        all_chunks = [np.frombuffer(x, np.float64) for x in co.allgather_bytes(chunks.tobytes())]
        # sum_w exactly as `for m in range(M): sum_w += Y[m].w` forms it (:411-413)
        self._exact_running_sums(1.0, all_chunks)
        lasts = [np.frombuffer(x, np.float64)[0] for x in co.allgather_bytes(np.float64(ctx.fs_shard_last_sum()).tobytes())]
        self.total = max(1e-8, float(lasts[-1]))                 # :415
        self._exact_running_sums(self.total, all_chunks)         # w /= sum_w (:418) and c_i (:421-428)
        # only the particles that some position draws travel (fast_slam1.py:421-431 reads Y[i] only for the drawn i;
        # with hundreds of landmarks the weights are peaked and a handful of ancestors feeds the whole next set)
        mine = ctx.shard_plan_drawn(self.firsts, rank, r)
        counts = np.stack([np.frombuffer(x, np.int64) for x in co.allgather_bytes(mine.tobytes())])    # [source, destination]
        send_counts = [int(c) for c in counts[rank]]
        recv_counts = [int(c) for c in counts[:, rank]]
        n_send, L = sum(send_counts), sum(recv_counts)
        self.rows_sent_total = getattr(self, "rows_sent_total", 0) + n_send
        self.rows_sent_off_rank = getattr(self, "rows_sent_off_rank", 0) + n_send - send_counts[rank]
        send = co.rows(n_send, 6)
        most_mine = 0
        if n_send:
            ctx.shard_pack_drawn(world, send.data_ptr())
            most_mine = ctx.fs_shard_apply_drawn()               # :318-404 for the drawn particles only
        # feature rows as long as the longest feature list that travels, not as the capacity
        most = max(int(np.frombuffer(x, np.int64)[0]) for x in co.allgather_bytes(np.int64(most_mine).tobytes()))
        ctx.fs_set_row_slots(max(1, min(most, self.capacity)))
        self.words = ctx.fs_row_words()
        fsend = co.rows(n_send, self.words, co.torch.int64)
        if n_send:
            ctx.fs_shard_pack_drawn(fsend.data_ptr())
        recv = co.all_to_all_rows(send, send_counts, recv_counts)
        frecv = co.all_to_all_rows(fsend, send_counts, recv_counts)
        self.last_idx = ctx.shard_unpack_drawn(recv.data_ptr(), L, r, True)
        ctx.fs_shard_unpack(frecv.data_ptr(), L)
        if not want_idx:
            self.last_idx = None
